// icpk_internal.h -- shared between the host side (icpk_api.cpp) and the HIP
// kernels.  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icpk {

// ---- K1 geometry ----------------------------------------------------------
// One query per lane, 256 lanes (4 wave64) per workgroup; the target cloud is
// streamed through LDS in tiles of NN_TILE points (3 planes x 4 KiB).  Target
// planes are padded to a multiple of NN_TILE with +inf (never selected).
constexpr int NN_THREADS = 256;
constexpr int NN_TILE = 1024;
static_assert(NN_TILE == 4 * NN_THREADS, "one float4 per lane per plane per tile");

// (distance bits << 32) | index: for non-negative floats the bit pattern is
// monotone, so an unsigned 64-bit min is the lexicographic (distance, index)
// min -- independent of the order target chunks are merged in.
typedef unsigned long long nn_key_t;
constexpr nn_key_t NN_KEY_INIT = ~0ull;

struct NnArgs {
  const float* qx;
  const float* qy;
  const float* qz;
  int nq;
  const float* tx;
  const float* ty;
  const float* tz;
  int nt_pad;           // multiple of NN_TILE
  int tiles_per_chunk;  // target tiles handled by one workgroup
  nn_key_t* best;       // [nq], pre-set to NN_KEY_INIT
  const int* stop;      // device-loop stop flags (LoopState), or nullptr
};

struct Rt {
  float R[9];
  float t[3];
};
struct LoopState;  // device-side loop state, defined below

constexpr int NSUM = 19;
constexpr int NSUM_REF = 13;  // sums [0..12]: all the reference flavour's loop step reads
constexpr int NP2L = 28;      // point-to-plane: 21 + 6 + 1
constexpr int NSUM_MAX = 28;
constexpr int RED_THREADS = 256;
constexpr int RED_MAX_BLOCKS = 256;

inline int red_blocks(int n) {
  int b = (n + RED_THREADS - 1) / RED_THREADS;
  if (b < 1) b = 1;
  if (b > RED_MAX_BLOCKS) b = RED_MAX_BLOCKS;
  return b;
}

// kernels_nn.hip
void launch_fill_u64(nn_key_t* p, int n, nn_key_t v, const int* stop, hipStream_t s);
void launch_nn_exact(const NnArgs& a, hipStream_t s);
// bounding boxes for the pruned scan: 6 planes (lo x,y,z, hi x,y,z) per 1024-target
// tile (tbox) and per 128-target sub-tile (sbox)
constexpr int NN_SUB = 128;
constexpr int NN_SUBS = NN_TILE / NN_SUB;  // 8 sub-tiles per tile
struct NnBoxes {
  float* tbox;
  int tbox_stride;  // floats per plane
  float* sbox;
  int sbox_stride;
  // pruned scan only: NnArgs::tx/ty/tz are the Morton-ordered planes the boxes were
  // built on; ox/oy/oz the target in the caller's order (seed look-up); tperm maps a
  // scanned position to the original index; qperm lists the queries in Morton order
  const float* ox;
  const float* oy;
  const float* oz;
  const int* tperm;
  const int* qperm;
};

// kernels_sort.hip
void launch_bounds(const float* tbox, int tbox_stride, int ntiles, float* bounds, hipStream_t s);
struct GridInfo;
void launch_morton_order(const float* x, const float* y, const float* z, int n, const float* bounds, int bits,
                         unsigned* keys, int* slot, int* count, int* start, int* bsum, GridInfo* table,
                         unsigned* keys_out, int* perm_out, hipStream_t s);
void launch_gather_planes(const float* x, const float* y, const float* z, const int* perm, int n, int n_pad, float pad,
                          float* ox, float* oy, float* oz, int* perm_pad, hipStream_t s);
void launch_nn_filtered(const NnArgs& a, const nn_key_t* seed, int seed_scale, int q_per_lane, hipStream_t s);
// kernels_nn_pruned.hip
// st != nullptr (device-side loop): K3 is fused into the sweep (see the kernel)
void launch_nn_pruned(const NnArgs& a, const nn_key_t* seed_m, nn_key_t* best_m, const NnBoxes& b, int slices,
                      int recheck, const LoopState* st, hipStream_t s);
void launch_seed_morton(const unsigned* qkeys, const int* qperm, int nq, const float* qx, const float* qy,
                        const float* qz, const unsigned* tkeys, const float* sx, const float* sy, const float* sz,
                        const int* tperm, int nt, nn_key_t* seed_m, hipStream_t s);
void launch_seed_gather(const nn_key_t* best, const int* qperm, int nq, nn_key_t* seed_m, hipStream_t s);
void launch_tile_boxes(const float* x, const float* y, const float* z, int n, int ntiles, const NnBoxes& b,
                       hipStream_t s);
void launch_decimate(const float* x, const float* y, const float* z, int n, int stride, float* ox, float* oy, float* oz,
                     int n_out_pad, hipStream_t s);
// kernels_grid.hip: uniform grid over the target (ICPK_NN_GRID)
#ifndef ICPK_GRID_MAX_CELLS_LOG2
#define ICPK_GRID_MAX_CELLS_LOG2 23  // 3 tables of 32 MB per context; 22 clipped the cell edge of the dense clouds
#endif
constexpr int GRID_MAX_CELLS = 1 << ICPK_GRID_MAX_CELLS_LOG2;
// the slots of the frame-batch mode (up to 32 child contexts) own smaller tables: Kinect-size pairs need ~0.9 M cells,
// and 3 x 8 MB per slot instead of 3 x 32 MB keeps a 64-pair batch at 0.8 GB instead of 3.2 GB (a denser pair in a slot
// merely gets a coarser grid: efficiency only)
constexpr int GRID_MAX_CELLS_SLOT = 1 << 21;
constexpr int GRID_BOUNDS_PARTS = 256;  // partial boxes of the bounds pass (6 floats each)
struct GridInfo {
  float lo[3];  // finite lower corner of the target
  float inv_h;  // 1 / cell edge (y and z)
  float h;
  float inv_hx;  // 1 / cell edge along x = xdiv / h
  int xdiv;
  int nx, ny, nz;
  int ncells;
  int nxq;       // cells along x of the coarser grid that orders the queries (nx / xdiv)
  int ncells_q;
};
void launch_grid_bounds(const float* x, const float* y, const float* z, int n, float* fb, hipStream_t s);
void launch_grid_info(const float* fb, int n, float ppc, int xdiv, int max_cells, GridInfo* g, hipStream_t s);
void launch_grid_tscatter(const float* x, const float* y, const float* z, const int* tcell, const int* tslot,
                          const int* cell_start, int n, float4* t4, float4* o4, hipStream_t s);
void launch_grid_qslot(const float* x, const float* y, const float* z, int n, const GridInfo* g, int* count, int* qcell,
                       int* qslot, int coarse, hipStream_t s);
// qm4 != nullptr: also the scan-order queries, element 0 as every query's seed point and seed key
void launch_grid_qscatter(const int* qcell, const int* qslot, const int* qstart, int n, int* qperm, const float* qx,
                          const float* qy, const float* qz, const float* ox, const float* oy, const float* oz,
                          float4* qm4, float4* sp, nn_key_t* seed_m, hipStream_t s);
constexpr int GRID_SCAN_BLOCKS = (GRID_MAX_CELLS + 1 + 2047) / 2048 + 1;  // scratch ints of launch_grid_scan
void launch_grid_scan(int* count, int* out, int* bsum, const GridInfo* g, int coarse, hipStream_t s);  // count[] left zero
// one pair's arguments of a grid sweep (K1d); see nn_grid_body
struct GridSweepArgs {
  float *qx, *qy, *qz;  // the caller's source planes (kept in step when K3 is fused)
  int nq;
  int pad0;
  float4* qm4;            // queries in scan order (x, y, z, original index)
  const float4* t4;       // targets sorted by cell (x, y, z, original index)
  const int* cell_start;
  const GridInfo* gi;
  const float *ox, *oy, *oz;  // target in the caller's order (literal seed, element 0)
  const float4* sp_in;    // seeds as points, scan order
  float4* sp_out;         // matches as points = the next sweep's seeds
  nn_key_t* best;         // results, caller's order
  nn_key_t* best_m;       // results, scan order
  const LoopState* st;    // device-side loop state (K3 fused) or nullptr
  float4* rec;            // device loop: caller-order records {(query, distance), (match, index)} INSTEAD of the planes / best / best_m stores; else nullptr
};
// frame-batch mode: up to BATCH_MAX independent pairs advance in lock step, one launch per
// stage for the whole group (blockIdx.y / blockIdx.x = pair); arguments travel by value.
constexpr int BATCH_MAX = 16;
struct GridSweepBatch {
  GridSweepArgs p[BATCH_MAX];
};
void launch_nn_grid(const GridSweepArgs& a, int slices, int expand, hipStream_t s);
void launch_nn_grid_batch(const GridSweepBatch& b, int count, int slices, int expand, hipStream_t s);
void launch_grid_unpack(const float4* qm4, const float4* rec, int nq, float* qx, float* qy, float* qz, nn_key_t* best,
                        hipStream_t s);
void launch_grid_query_points(const float* qx, const float* qy, const float* qz, const int* qperm, int nq,
                              const nn_key_t* seed_m, const float* ox, const float* oy, const float* oz, float4* qm4,
                              float4* sp, hipStream_t s);
constexpr int NN_SEED_STRIDE = 16;  // decimation of the target for the seeding pre-pass
void launch_pair_distance(const float* a, const float* b, float* out, int n, int point3, hipStream_t s);

// ---- device-side ICP loop (kernels_loop.hip) ------------------------------------
// One LoopState per context lives in device memory; while an alignment runs, every
// kernel of every iteration is enqueued up front and consults it: once `done` (or
// `stop_after_transform`) is set the remaining launches are no-ops, so no host
// round trip is needed to decide the loop exit of icp.cpp:155.
constexpr int LOOP_MAX_ITER = 256;
struct LoopState {
  int done;                  // loop exited (threshold met / max iterations / degenerate)
  int stop_after_transform;  // < min_pairs fallback (icp.cpp:163-182): apply rt, then stop
  int iterations;            // completed loop bodies
  int status;
  int sweeps;                // NN sweeps executed (tells the host which buffer holds the result)
  int steps;                 // loop steps executed (mirrored to *progress)
  long long pairs;           // associations of the latest sweep
  float mse;                 // icp.cpp:622-638 of the latest sweep
  int epoch;                 // tag of this alignment in the progress words (see below)
  // host-visible (pinned, mapped) progress words or nullptr: [0] = loop steps executed, [1] = loop has
  // exited.  With a loop that may exit early (threshold mode) the host enqueues only a couple of
  // iterations ahead of them instead of all max_iterations: every launch after the exit is a no-op that
  // still costs its dispatch (~3 us each: 100 us for the reference's 16 / 1e-4 setting leaving after 5).
  // Both words carry `epoch` ([0] = epoch << 10 | steps, [1] = epoch << 2 | done << 1 | exited), so that words still
  // being written by an alignment that was abandoned on an error are never taken for this one's.
  int* progress;
  // host-visible (pinned, mapped) copy of the OUTPUT fields of this struct or nullptr: the step that ends the loop
  // writes them there and then sets progress[2] = epoch << 1 | 1 with a system-scope release, so the host has the
  // result the moment the loop ends -- no copy kernel, no stream wait (trace entries go there as they are made)
  LoopState* mirror;
  Rt rt;                     // transform to apply in this iteration
  double Rd[9];              // rt.R widened (exact) by the step: the sweeps' fused K3 takes the rotation as float64 scalars
  float Trot[9];             // icp.cpp:227-233
  float offset[3];           // icp.cpp:240
  double Tk[12];             // accumulated [R|t] (Kabsch / point-to-plane)
  // parameters
  int max_iterations, min_pairs, solve, fixed_iterations;
  float threshold;
  float last_rotation[9], last_translation[3];
  // per-iteration record (icpk_get_trace)
  float trace_R[LOOP_MAX_ITER * 9];
  float trace_t[LOOP_MAX_ITER * 3];
  float trace_mse[LOOP_MAX_ITER];
  int trace_pairs[LOOP_MAX_ITER];
};
// the first two ints of LoopState, as seen by kernels that only need to know whether to run
__device__ __forceinline__ bool loop_stopped(const int* stop) { return stop && (stop[0] | stop[1]); }

// sums the per-block partials with the canonical tree and, unless stats_only, performs
// one loop body's host work on the device: exit test, solve, pose accumulation, trace
struct StepArgs {
  const double* partial;
  const int* pcount;
  int nblocks;
  int pad0;
  LoopState* st;
};
struct StepBatch {
  StepArgs p[BATCH_MAX];
};
void launch_loop_step(const double* partial, const int* pcount, int nblocks, int nsum, LoopState* st, int stats_only,
                      hipStream_t s);
void launch_loop_step_batch(const StepBatch& b, int count, int nsum, int stats_only, hipStream_t s);
void launch_transform_state(float* x, float* y, float* z, int n, const LoopState* st, hipStream_t s);
void launch_reduce_final(const double* partial, const int* pcount, int nblocks, int nsum, double* out, hipStream_t s);
// query-sharded loop: NSUM sums + the count as a double into out[0 .. NSUM]; launch_loop_step with nblocks = -1 then
// takes the (all-reduced) sums from there
void launch_reduce_final_shard(const double* partial, const int* pcount, int nblocks, double* out, const LoopState* st,
                               hipStream_t s);

// kernels_reduce.hip
// partial: [NSUM_MAX][RED_MAX_BLOCKS] doubles (sum-major: stage 2 reads it coalesced), pcount: [RED_MAX_BLOCKS] ints,
// out: nsum doubles followed by one int64 count ((NSUM_MAX + 1) x 8 bytes).
// out == nullptr: only the per-block partials are produced (the device loop sums them
// in launch_loop_step); stop: device-loop stop flags or nullptr
// o4: the target as caller-order (x, y, z, 0) points or nullptr (then tx / ty / tz are gathered)
// rec != nullptr (device loop behind a grid sweep): everything comes from the sweep's caller-order records instead
void launch_assoc_reduce(const nn_key_t* best, const float* ax, const float* ay, const float* az, int nq,
                         const float* tx, const float* ty, const float* tz, const float4* o4, const float4* rec, float max_dist,
                         int32_t* idx_out, float* dist_out, double* partial, int* pcount, double* out, LoopState* st,
                         int nact, hipStream_t s);
// one pair's arguments of K2 inside a device loop (no idx/dist unpacking, no final stage)
struct ReduceArgs {
  const nn_key_t* best;
  const float *ax, *ay, *az;
  const float *tx, *ty, *tz;
  const float4* o4;  // or nullptr
  const float4* rec;  // the sweep's caller-order records
  double* partial;
  int* pcount;
  LoopState* st;
  int nq;
  int nblocks;  // red_blocks(nq): the canonical geometry of THIS pair
};
struct ReduceBatch {
  ReduceArgs p[BATCH_MAX];
};
void launch_assoc_reduce_batch(const ReduceBatch& b, int count, float max_dist, int nact, hipStream_t s);

void launch_p2l_reduce(const nn_key_t* best, const float* ax, const float* ay, const float* az, int nq, const float* tx,
                       const float* ty, const float* tz, const float* nx, const float* ny, const float* nz,
                       const float4* rec, float max_dist, int32_t* idx_out, float* dist_out, double* partial, int* pcount,
                       double* out, LoopState* st, hipStream_t s);

// kernels_transform.hip
void launch_transform(float* x, float* y, float* z, int n, const Rt& rt, hipStream_t s);
void launch_fill_f32(float* p, int n, float v, hipStream_t s);
void launch_ingest_cloud(const float* x, const float* y, const float* z, int n, int n_pad, float pad, float* d1, int cap1,
                         float* d2, int cap2, hipStream_t s);

// kernels_backproject.hip
// counts: [ceil(npix/1024)+1] ints scratch.  Returns nothing; *n_out (device)
// receives the number of points.
// nx == nullptr: no normals
void launch_backproject(const uint16_t* depth, int rows, int cols, float fx, float cx, float ox, float oy, float oz,
                        float* x, float* y, float* z, float* nx, float* ny, float* nz, int normals_mode,
                        int* block_counts, int* n_out, unsigned long long sub_key, int sub_factor, hipStream_t s);
// icpk_backproject_pair: image 0 = current frame (source; x2/y2/z2 = its working copy), image 1 = previous
// frame (target).  counts: nblocks + 2 ints per image.
struct BpImage {
  const uint16_t* depth;
  float *x, *y, *z;
  float *x2, *y2, *z2;
  int* counts;
  float pad;
  // image-space seeds of the alignment that follows (see QscatterArgs::spix) or nullptr: the pixel of every point
  // (per point) / the point of every pixel, -1 where the pixel is empty (per pixel)
  int* pixel_of_point;
  int* point_of_pixel;
  // zero-copy upload or nullptr: the image still sits in pinned host memory; the counting pass reads it from there
  // (one PCIe read per pixel, no copy-engine command in front of the kernels) and leaves the device copy in raw_out,
  // which `depth` points to for the scatter pass
  const uint16_t* host_src;
  uint16_t* raw_out;
  unsigned long long sub_key;  // seeded subsample of this image (kernels_backproject.hip: bp_keep); factor <= 1: none
  int sub_factor, pad1;
};
struct BpPair {
  BpImage im[2];
};
void launch_backproject_pair(const BpPair& b, int rows, int cols, float fx, float cx, float ox, float oy, float oz,
                             const Rt& rt, int posed, int* n_out, int* n_host, hipStream_t s);

// kernels_frontend.hip
// order-preserving split of a sweep's result into accepted pairs and rejected queries
// (icp.cpp:488-515); block_counts: ceil(nq/1024) + 1 ints of scratch, *n_accepted (device) = count
void launch_assoc_split(const nn_key_t* best, int nq, float max_dist, int* block_counts, int* n_accepted,
                        int32_t* assoc_q, int32_t* assoc_t, float* assoc_d, int32_t* rej_q, hipStream_t s);
// SLAM.cpp:553-574: range clamp, then (morph != 0) 5x5 dilate + erode with anchor (ax, ay)
void launch_depth_filter(const uint16_t* in, uint16_t* out, int rows, int cols, int min_d, int max_d, int ax, int ay,
                         int morph, hipStream_t s);


// ---- deferred set-up launches of the frame-batch mode ------------------------------------------------
// A lock-step group's pairs go through the SAME sequence of small set-up kernels (ingest x 2, loop-state
// init, bounds, grid info, 2 x (cell slots, scan sums, scan), target scatter, query scatter): 13 launches per
// pair on its own stream, which for 8 pairs is 0.26 ms that nothing hides when the group is the only one
// (config 4 at 8 GPUs).  While a SetupRecorder is installed in the calling thread the launchers below
// RECORD their arguments instead of launching; flush_setup_batches() then issues ONE launch per step for
// all pairs (blockIdx.y / blockIdx.x = pair, arguments by value), or, should the pairs' sequences differ,
// replays them one by one.  Same kernels' bodies: same bits.
struct IngestArgs {
  const float *x, *y, *z;
  int n, n_pad;
  float pad;
  int cap1;
  float* d1;
  float* d2;
  int cap2;
  int pad0;
};
struct LoopInitArgs {
  LoopState* st;
  int* progress;
  LoopState* mirror;
  int max_iterations, min_pairs, solve, fixed_iterations;
  float threshold;
  int epoch;
  float last_rotation[9], last_translation[3];
};
struct BoundsArgs {
  const float *x, *y, *z;
  float* fb;
  int n, nparts;
};
struct InfoArgs {
  const float* fb;
  GridInfo* g;
  int nparts, n;
  float ppc;
  int xdiv;
  int max_cells;  // capacity of this context's cell tables
  int pad0;
};
struct QslotArgs {
  const float *x, *y, *z;
  const GridInfo* gi;
  int *count, *cell, *slot;
  int n, coarse;
};
struct ScanArgs {
  int *count, *out, *bsum;
  const GridInfo* g;
  int coarse, pad0;
};
struct TscatterArgs {
  const float *x, *y, *z;
  const int *tcell, *tslot, *cell_start;
  float4 *t4, *o4;
  int n, pad0;
};
struct QscatterArgs {
  const int *qcell, *qslot, *qstart;
  int* qperm;
  const float *qx, *qy, *qz, *ox, *oy, *oz;
  float4 *qm4, *sp;
  nn_key_t* seed_m;
  int n, pad0;
  // Seeds from the images the two clouds were back-projected from (icpk_backproject_pair), or nullptr: spix[i] = pixel
  // of query i, tidx[p] = target point of pixel p or -1.  Consecutive frames of a depth camera put the same surface
  // within a pixel or two, so the target point of the query's own pixel (or of the nearest occupied pixel of the
  // 5 x 5 around it) is a seed centimetres from the true match -- instead of the reference's literal element 0, metres
  // away (icp.cpp:572).  A seed only sets where the search starts: the result is the exact nearest neighbour either way.
  const int* spix;
  const int* tidx;
  int rows, cols;
};
enum SetupKind : int { SK_INGEST, SK_LOOP_INIT, SK_BOUNDS, SK_INFO, SK_QSLOT, SK_SCAN, SK_TSCATTER, SK_QSCATTER };
struct SetupCall {
  int kind;
  union {
    IngestArgs ingest;
    LoopInitArgs loop_init;
    BoundsArgs bounds;
    InfoArgs info;
    QslotArgs qslot;
    ScanArgs scan;
    TscatterArgs tscatter;
    QscatterArgs qscatter;
  };
};
// a pair's set-up is 13 steps today (ingest x 2, loop init, bounds, info, 2 x (slots, scan), target scatter, query
// scatter + the upload path's extras); a recorder that overflows marks itself and the pair FAILS (align_batch_impl)
constexpr int SETUP_MAX_CALLS = 24;
struct SetupRecorder {
  SetupCall calls[SETUP_MAX_CALLS];
  int n = 0;
  bool overflow = false;
};
SetupRecorder*& setup_recorder();  // of the calling thread; nullptr: launch at once
template <typename A>
struct SetupBatchOf {
  A p[BATCH_MAX];
};
void launch_grid_begin(const float* x, const float* y, const float* z, int n, float* fb, float ppc, int xdiv, int max_cells,
                       GridInfo* g, const LoopInitArgs* init, int* ticket, hipStream_t s);
void launch_loop_init(const LoopInitArgs& a, hipStream_t s);
// one launch per recorded step for `count` pairs, in recording order; returns false if the sequences differ
// (nothing launched then: replay them with replay_setup)
bool flush_setup_batches(const SetupRecorder* recs, int count, hipStream_t s);
void replay_setup(const SetupRecorder& rec, hipStream_t s);
void launch_ingest_batch(const SetupBatchOf<IngestArgs>& b, int count, hipStream_t s);
void launch_loop_init_batch(const SetupBatchOf<LoopInitArgs>& b, int count, hipStream_t s);
void launch_grid_bounds_batch(const SetupBatchOf<BoundsArgs>& b, int count, hipStream_t s);
void launch_grid_info_batch(const SetupBatchOf<InfoArgs>& b, int count, hipStream_t s);
void launch_grid_qslot_batch(const SetupBatchOf<QslotArgs>& b, int count, hipStream_t s);
void launch_grid_scan_batch(const SetupBatchOf<ScanArgs>& b, int count, hipStream_t s);
void launch_grid_tscatter_batch(const SetupBatchOf<TscatterArgs>& b, int count, hipStream_t s);
void launch_grid_qscatter_batch(const SetupBatchOf<QscatterArgs>& b, int count, hipStream_t s);
void launch_grid_tqscatter(const TscatterArgs& t, const QscatterArgs& q, hipStream_t s);

}  // namespace icpk
