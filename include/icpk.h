/*
 * icpk.h -- C ABI of libicpk.so: the MI355X (gfx950) implementation of the ICP
 * inner loop of BenniG123/icp-slam-prototype.
 *
 * This is the drop-in boundary.  The reference has no
 * FFI: its hot path is C++ called in-process.  Each entry point below names the
 * reference code it replaces ("file:line" relative to the reference checkout).
 * Plain pointers and sizes only; no C++/torch types; nothing throws across the
 * boundary.  All functions return an int status (ICPK_OK == 0, negative =
 * error, positive = completed with a documented fallback) unless noted.
 *
 * There is NO CPU fallback: icpk_create fails with ICPK_E_NO_DEVICE when no HIP
 * device is usable.
 *
 * Data layout: point clouds are xyz structure-of-arrays (three float planes),
 * replacing the reference's 16-byte AoS color_point_t (pointcloud.hpp:13-19);
 * colour is dropped because COLOR_WEIGHT is 0.0f (icp.hpp:6).  Coordinates are
 * expected to be finite (the reference produces them from uint16 depth): a NaN/inf
 * point never faults or hangs a kernel and never pairs, but which index it reports
 * is unspecified.
 */
#ifndef ICPK_H
#define ICPK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICPK_VERSION_STRING "icpk 0.3.0 (gfx950)"

/* ---- status codes -------------------------------------------------------- */
#define ICPK_OK 0
#define ICPK_W_DEGENERATE 2       /* point-to-plane normal equations not positive definite:  \
                                    iteration stopped, transform so far returned            */
#define ICPK_W_EMPTY_MAP 3       /* icpk_associate_keypoints with an empty target: outputs left \
                                    untouched, as icp.cpp:490-491 returns before clearing them  */
#define ICPK_W_TOO_FEW_PAIRS 1   /* < min_pairs associations: fell back to the  \
                                    caller's last motion (icp.cpp:163-182)      */
#define ICPK_E_ARG (-1)          /* null pointer / negative size / bad enum     */
#define ICPK_E_EMPTY_TARGET (-2) /* icp.cpp:572 dereferences begin(): UB there  */
#define ICPK_E_HIP (-3)          /* a HIP runtime call failed (see last_error)  */
#define ICPK_E_NOT_SET (-4)      /* source or target not uploaded yet           */
#define ICPK_E_NO_DEVICE (-5)    /* no usable HIP device: no CPU fallback       */
#define ICPK_E_RCCL (-6)         /* librccl missing or an RCCL call failed       */

/* ---- reference constants (defaults of icpk_default_params) --------------- */
#define ICPK_MAX_NN_DISTANCE 0.75f          /* icp.hpp:8  MAX_NN_COLOR_DISTANCE    */
#define ICPK_MAX_NN_KEYPOINT_DISTANCE 0.1f  /* icp.hpp:10 MAX_NN_KEYPOINT_DISTANCE */
#define ICPK_DEFAULT_MAX_ITERATIONS 16      /* SLAM.cpp:277                        */
#define ICPK_DEFAULT_THRESHOLD 0.0001f      /* SLAM.cpp:277                        */
#define ICPK_MIN_PAIRS 3                    /* icp.cpp:163                         */
#define ICPK_FX 468.60f                     /* pointcloud.hpp:7                    */
#define ICPK_CX 318.27f                     /* pointcloud.hpp:9                    */
#define ICPK_DEPTH_SCALE 5000.0f            /* pointcloud.cpp:37                   */

/* solve flavours */
#define ICPK_SOLVE_REFERENCE 0 /* bug-for-bug icp.cpp:199-246 (un-centred moment,    \
                                  R = V U^T, column-2 flip, mean-difference offset)  */
#define ICPK_SOLVE_KABSCH 1    /* centred Kabsch, rigid_transform_3D.py:9-40         */
#define ICPK_SOLVE_POINT_TO_PLANE 2 /* linearised point-to-plane (extension: TODO:9 of the \
                                       reference only plans it); needs target normals     */

/* nearest-neighbour kernel selection */
#define ICPK_NN_EXACT 0    /* literal double-precision distance per pair             */
#define ICPK_NN_FILTERED 1 /* seeded fp32 filter + exact re-evaluation; same results */
#define ICPK_NN_PRUNED 2   /* FILTERED + skipping of target tiles whose bounding box is out \
                              of reach; same results                                   */
#define ICPK_NN_GRID 3     /* uniform grid over the target: only the cells that meet the cube \
                              [q - r, q + r] around a query with seed distance r are scanned; \
                              same results; default                                     */

/* log keys mirrored from SLAM.hpp:4-13 for the optional callback */
#define ICPK_LOG_NEAREST_NEIGHBOR 0
#define ICPK_LOG_RECONSTRUCT_POINT_CLOUDS 5
#define ICPK_LOG_SVD 6
#define ICPK_LOG_ROTATE 7
#define ICPK_LOG_MSE 8

/* Canonical reduction geometry (part of the ABI: it fixes the summation order
 * of icpk_reduce so results are bit-reproducible and checkable): 256-thread
 * blocks, B = clamp(ceil(n/256), 1, 256) blocks, thread g sums elements
 * g, g+256B, ... in order; 64-lane xor butterfly; ((w0+w1)+w2)+w3; the B block
 * sums (padded to 256 slots with +0.0) go through the same 4-wave tree again. */
#define ICPK_RED_THREADS 256
#define ICPK_RED_MAX_BLOCKS 256
#define ICPK_NP2L 28 /* point-to-plane sums: [0..20] upper triangle of J J^T (row-major),  \
                        [21..26] J r, [27] sum dist; J = [p x n ; n], r = (p - q).n      */
#define ICPK_NSUM 19 /* [0..8] M[r][c]=sum b_r a_c, [9..11] sum (float)(a-b), \
                        [12] sum dist, [13..15] sum a, [16..18] sum b          */

typedef struct icpk_ctx icpk_ctx; /* opaque: owns device buffers + one HIP stream */

typedef struct icpk_params {
  int32_t max_iterations;   /* SLAM.cpp:277 (16)                                  */
  float threshold;          /* SLAM.cpp:277 (1e-4): loop while mse > threshold    */
  float max_nn_dist;        /* icp.hpp:8 (0.75) or icp.hpp:10 (0.1)               */
  int32_t min_pairs;        /* icp.cpp:163 (3)                                    */
  int32_t solve;            /* ICPK_SOLVE_*                                       */
  int32_t fixed_iterations; /* 1: ignore threshold, run max_iterations (bench)    */
  int32_t nn_mode;          /* ICPK_NN_*                                          */
  int32_t profile;          /* 1: HIP events around the NN kernels -> stats.nn_ms_total;
                               2: around every stage (reduce, transform) as well   */
  float last_rotation[9];    /* caller's previous motion, icp.cpp:23,176          */
  float last_translation[3]; /* icp.cpp:25,177                                    */
  int32_t host_loop;        /* 0 (default): every iteration's kernels are enqueued up front
                               and the loop test / solve run on the device (no host round
                               trip per iteration); 1: the host drives each iteration and
                               solves (one 160-byte read-back per iteration).  Same results. */
  int32_t profile_stride;   /* profile == 1 only: bracket every n-th NN launch (0, 1: every one;
                               the offset advances with every alignment, so the sample covers all
                               sweep positions); an event pair costs ~4 us of queue time, which a
                               20 us kernel notices */
} icpk_params;

typedef struct icpk_stats {
  int32_t iterations;  /* completed loop bodies (icp.cpp:257)                     */
  int32_t status;      /* same value icpk_align returned                          */
  int32_t final_pairs; /* associations after the last sweep                       */
  float final_mse;     /* meanSquareError of the last sweep (icp.cpp:264)         */
  int32_t nn_launches; /* NN sweeps launched (= iterations + 1)                   */
  int32_t nn_timed_launches; /* NN launches bracketed by events (see profile_stride)    */
  /* device times measured with HIP events on the context's stream; filled only
   * when params.profile != 0 */
  float nn_ms_total;        /* sum over the nn_timed_launches bracketed NN kernels */
  float reduce_ms_total;    /* association reduce kernels                         */
  float transform_ms_total; /* point transform kernels                            */
  float total_ms;           /* first to last recorded event                       */
} icpk_stats;

/* One frame pair for icpk_align_batch (xyz-SoA: host pointers; icpk_align_batch_device:
 * device pointers on the context's device).  idx_out / dist_out: optional HOST arrays of ns
 * entries that receive the pair's final associations (as icpk_get_associations); NULL = not
 * wanted. */
typedef struct icpk_pair {
  const float *sx, *sy, *sz;
  int32_t ns;
  const float *tx, *ty, *tz;
  int32_t nt;
  int32_t *idx_out;
  float *dist_out;
} icpk_pair;

/* same shape as logDeltaTime(int logKey, int quantity) (SLAM.hpp:30,
 * SLAM.cpp:493-510) plus the elapsed microseconds the reference computes
 * internally */
typedef void (*icpk_log_fn)(int key, int quantity, double usec, void *user);

/* ---- lifetime ------------------------------------------------------------ */
const char *icpk_version(void);
int icpk_create(icpk_ctx **out, int device_id);
void icpk_destroy(icpk_ctx *ctx);
const char *icpk_last_error(const icpk_ctx *ctx);
void icpk_default_params(icpk_params *p);
int icpk_set_log_callback(icpk_ctx *ctx, icpk_log_fn fn, void *user);
/* the HIP stream all work of this context is enqueued on (hipStream_t) */
void *icpk_stream(icpk_ctx *ctx);

/* ---- clouds: replaces the PointCloud containers built at icp.cpp:38-39 ---- */
/* host pointers (copied; caller keeps ownership) */
int icpk_set_target(icpk_ctx *ctx, const float *x, const float *y, const float *z, int32_t n);
int icpk_set_source(icpk_ctx *ctx, const float *x, const float *y, const float *z, int32_t n);
/* device pointers on the context's device (copied device-to-device) */
int icpk_set_target_device(icpk_ctx *ctx, const float *dx, const float *dy, const float *dz, int32_t n);
int icpk_set_source_device(icpk_ctx *ctx, const float *dx, const float *dy, const float *dz, int32_t n);
/* working copy of the source <- the cloud last given to icpk_set_source*     */
int icpk_reset_source(icpk_ctx *ctx);
/* the cloud icpk_align / icpk_reset_source start from <- the working copy (makes
 * transforms applied with icpk_transform_source permanent; device-side copy) */
int icpk_commit_source(icpk_ctx *ctx);
/* current (transformed) source, to host */
int icpk_get_source(icpk_ctx *ctx, float *x, float *y, float *z);
/* target cloud as the device holds it (after icpk_transform_target / icpk_backproject) */
int icpk_get_target(icpk_ctx *ctx, float *x, float *y, float *z);
int32_t icpk_source_size(const icpk_ctx *ctx);
int32_t icpk_target_size(const icpk_ctx *ctx);

/* ---- the three steps of one iteration ------------------------------------ */
/* icp.cpp:541-563 findGlobalNearestNeighborAssociations + :566-593
 * getNearestPoint + :606-620 distance.  For every source point the index of
 * the target element the reference scan would copy (strict '<' on the float
 * distance, lowest index wins ties) and that distance.  Outputs may be NULL
 * (results stay on the device for icpk_reduce).  The `d < max` acceptance of
 * icp.cpp:553 is applied by the consumers, not here. */
int icpk_nn(icpk_ctx *ctx, int32_t nn_mode, int32_t *idx_out, float *dist_out);
/* icp.cpp:186-212 (association split + cross moment), :314-344
 * (calculateOffset sums), :622-638 (meanSquareError sum), in one pass over the
 * associations of the last icpk_nn, canonical order.  sums: ICPK_NSUM doubles. */
int icpk_reduce(icpk_ctx *ctx, float max_dist, double *sums, int64_t *count);
/* pointcloud.cpp:321-346 rotate + :349-359 translate:
 * p <- fl32(fl32(R p) + t), R row-major, applied to the working source. */
int icpk_transform_source(icpk_ctx *ctx, const float R[9], const float t[3]);
/* same transform applied to the target cloud (the reference moves the previous
 * frame into the world frame the same way, icp.cpp:58-59) */
int icpk_transform_target(icpk_ctx *ctx, const float R[9], const float t[3]);
/* associations of the last sweep (device -> host) */
int icpk_get_associations(icpk_ctx *ctx, int32_t *idx_out, float *dist_out);

/* icp.cpp:488-515 findGlobalKeyPointAssociations + :517-539 getNearestKeyPoint -- the LIVE
 * association of the reference (icp.cpp:98,255) -- on the context's clouds: source = the frame's
 * key points, target = the map's key points.  One NN sweep (same kernels as icpk_nn), then the
 * order-preserving split on the device:
 *   assoc_query / assoc_target / assoc_dist  [*n_assoc]  accepted pairs in query order
 *       (`associations`, `errors`; rebuilt by every call, icp.cpp:497-498), accepted iff
 *       dist < max_dist (MAX_NN_KEYPOINT_DISTANCE 0.1f, icp.hpp:10 / icp.cpp:503);
 *   rejected_query [rejected_capacity]  the rejected query indices are APPENDED at *n_rejected,
 *       which is in/out (`nonAssociations` is never cleared between sweeps, icp.cpp:507-509).
 * Arrays need room for the source size.  Empty target: returns ICPK_W_EMPTY_MAP and touches
 * nothing (icp.cpp:490-491).  Inside icpk_align the same acceptance is params.max_nn_dist. */
int icpk_associate_keypoints(icpk_ctx *ctx, int32_t nn_mode, float max_dist, int32_t *assoc_query,
                             int32_t *assoc_target, float *assoc_dist, int32_t *n_assoc,
                             int32_t *rejected_query, int32_t rejected_capacity, int32_t *n_rejected);

/* ---- whole loop: replaces icp.cpp:98-268 --------------------------------- */
/* Starts from the source as uploaded (icpk_reset_source), leaves the aligned
 * source on the device.  T_out: row-major 4x4, same content as the CV_32FC1
 * matrix icp::getTransformation returns (icp.cpp:29,227-233,266-268) with row
 * 3 = (0,0,0,1) instead of uninitialised memory.  stats may be NULL.
 * Returns as soon as T_out / stats / the trace are final (the device writes them into pinned, mapped host memory);
 * the last launches of the call -- the caller-order copy of the aligned source among them -- may still be running,
 * and every later call on this context is ordered behind them (icpk_get_source and icpk_get_associations wait). */
int icpk_align(icpk_ctx *ctx, const icpk_params *p, float T_out[16], icpk_stats *stats);
/* Per-iteration record of the last icpk_align: for iteration i < *n_iter,
 * R_out[9*i..] is the rotation found (icp.cpp:218-223, before inversion),
 * t_out[3*i..] the offset (reference flavour, icp.cpp:240) or translation
 * (Kabsch), pairs_out[i] / mse_out[i] the association count and MSE the loop
 * test at icp.cpp:155 saw.  The caller needs these to keep the reference's pose
 * state (cameraRotation *= R^-1, cameraPosition -= offset, icp.cpp:237,246).
 * Arrays sized for params.max_iterations entries; any may be NULL. */
int icpk_get_trace(icpk_ctx *ctx, int32_t *n_iter, float *R_out, float *t_out, int32_t *pairs_out,
                   float *mse_out);
/* frame-batch mode (the frame-pair formulation of icp.cpp:541-563): n_pairs
 * independent pairs on this context's device; T_out n_pairs x 16, stats n_pairs (or NULL).
 * With the default kernels (ICPK_NN_GRID, device-side loop, reference or Kabsch flavour) up to
 * ICPK_BATCH_GROUP (default and at most 16) pairs advance in lock step -- one launch per stage
 * for the whole group -- while the next group is being uploaded and indexed; every pair's
 * result equals icpk_align on that pair bit for bit.  Other settings run the pairs one after
 * the other.  The context's own clouds are not touched by the lock-step path.  params.profile = 1
 * in this mode: ONE batched NN launch per group (its position rotating) is bracketed by HIP events;
 * the time is booked on the group's first pair (stats.nn_ms_total, nn_timed_launches = 1) and covers
 * ALL pairs of that group.
 * Returns the first negative status, else the max status. */
int icpk_align_batch(icpk_ctx *ctx, int32_t n_pairs, const icpk_pair *pairs,
                     const icpk_params *p, float *T_out, icpk_stats *stats);
/* same with the clouds already resident in HBM (sx..tz are device pointers; copied
 * device-to-device into the slots, so the caller's buffers stay untouched) */
int icpk_align_batch_device(icpk_ctx *ctx, int32_t n_pairs, const icpk_pair *pairs,
                            const icpk_params *p, float *T_out, icpk_stats *stats);

/* ---- multi-GPU: RCCL over xGMI behind the C ABI ------- */
/* One process (or host thread) and one context per GPU.  The path shards over independent
 * frame pairs (frame-pair formulation of icp.cpp:541-563): no per-iteration collective; the
 * collectives are one broadcast of a shared target cloud (key frame) and one all-gather of the
 * results.  librccl.so.1 is opened on first use (dlopen), not linked. */
#define ICPK_COMM_ID_BYTES 128
/* rank 0 makes the id (ncclGetUniqueId) and ships it to the other ranks by the host's own
 * means (socket, file, MPI, torch store) */
int icpk_comm_unique_id(void *id_out /* ICPK_COMM_ID_BYTES */);
int icpk_comm_init_rccl(icpk_ctx *ctx, const void *unique_id, int rank, int world);
int icpk_comm_destroy(icpk_ctx *ctx);
int icpk_comm_rank(const icpk_ctx *ctx);   /* -1 without a communicator */
int icpk_comm_world(const icpk_ctx *ctx);  /*  0 without a communicator */
/* block-wise shard of n_items over world ranks: rank's contiguous [start, start + count) */
void icpk_comm_partition(int32_t n_items, int world, int rank, int32_t *start, int32_t *count);
/* the root's target cloud (icpk_set_target* / icpk_backproject there) becomes the target of
 * every rank: ncclBroadcast of the three planes, 3 * Nt * 4 bytes */
int icpk_comm_broadcast_target(icpk_ctx *ctx, int root);
/* results of a block-partitioned batch of n_total pairs: this rank contributes the n_local
 * rows of its block (T_local n_local x 16, stats_local n_local or NULL) and receives all rows
 * in global pair order: T_all n_total x 16, stats_all n_total x 4 four-byte slots (iterations, status,
 * final_pairs as int32 BIT PATTERNS -- memcpy them out, exact whatever the cloud size -- and final_mse as a
 * float) or NULL.  One ncclAllGather. */
int icpk_comm_gather_results(icpk_ctx *ctx, const float *T_local, const icpk_stats *stats_local,
                             int32_t n_local, int32_t n_total, float *T_all, float *stats_all);
/* query-sharded single pair (SURVEY.md 8e alternative): sums[0..n) and *count summed over the
 * ranks in place; one ncclAllReduce of (n + 1) doubles per iteration */
int icpk_comm_allreduce_sums(icpk_ctx *ctx, double *sums, int32_t n, int64_t *count);
int icpk_comm_barrier(icpk_ctx *ctx);
/* Query-sharded alignment of ONE pair over the communicator's ranks (SURVEY.md 8e alternative; the frame-pair
 * formulation icp.cpp:541-563 with the queries split): same target on every rank (icpk_comm_broadcast_target),
 * source = this rank's slice of the queries; collective call.  The whole loop is enqueued: per iteration the grid
 * sweep and the reduction on the slice, then ONE in-stream ncclAllReduce of 20 doubles (19 sums + pair count) and
 * the replicated loop step on the reduced sums -- no host round trip per iteration.  reference and Kabsch
 * flavours.  Every rank returns the same T and statistics (final_pairs = all ranks' pairs); they agree with
 * icpk_align on the whole pair to ~1e-6 (the ranks' sums are added in the collective's order), bit for bit
 * with one rank. */
int icpk_align_query_sharded(icpk_ctx *ctx, const icpk_params *p, float T_out[16], icpk_stats *stats);

/* ---- front end -------------------------------------- */
/* pointcloud.cpp:27-30: the reference keeps one valid pixel in SUBSAMPLE_FACTOR, chosen by an unseeded rand().
 * factor > 1: every back-projection of this context (icpk_backproject*, both images of icpk_backproject_pair --
 * the source first, then the target, as icp.cpp:38-39 builds them; a resident previous frame draws a fresh
 * pattern too, as the reference's rebuilt cloud does) keeps a valid pixel p iff
 *   z = seed + (k + 1) * 0x9E3779B97F4A7C15 + p * 0xD1B54A32D192ED03          (k = images since this call, mod 2^64)
 *   z = (z ^ z >> 30) * 0xBF58476D1CE4E5B9;  z = (z ^ z >> 27) * 0x94D049BB133111EB;  z ^= z >> 31
 *   (uint32)(z >> 32) % factor == 0
 * -- reproducible, unlike rand(); the stream of the reference's C library is not pinned by anything it ships.
 * factor 0 or 1 (the default): every valid pixel.  Resets k to 0. */
/* Optional, for callers whose frames live in long-lived buffers (a camera driver's ring, a reused cv::Mat): pins
 * [ptr, ptr + bytes) and maps it for the device (hipHostRegister).  icpk_backproject_pair then reads a depth image that
 * lies inside a registered range where it is -- no staging copy, no transfer command; the image has been consumed
 * when the call returns, as always.  The memory must stay allocated until icpk_unregister_host_buffer (or
 * icpk_destroy).  Not needed for correctness; results are the same. */
int icpk_register_host_buffer(icpk_ctx *ctx, const void *ptr, size_t bytes);
int icpk_unregister_host_buffer(icpk_ctx *ctx, const void *ptr);

#define ICPK_SUBSAMPLE_FACTOR 40 /* pointcloud.hpp:11 */
int icpk_set_subsample(icpk_ctx *ctx, int32_t factor, uint64_t seed);

/* pointcloud.cpp:19-58 (the subsample of :27-30 as set by icpk_set_subsample; none by default): row-major back-
 * projection of a rows x cols uint16 depth image (host pointer) into the
 * source (which = 0) or target (which = 1) cloud, adding `offset` to every
 * coordinate afterwards (PointCloud::translate(cameraPosition), icp.cpp:71).
 * Returns the number of points (>= 0) or a negative status. */
int icpk_backproject(icpk_ctx *ctx, const uint16_t *depth, int32_t rows, int32_t cols,
                     float fx, float cx, const float offset[3], int32_t which);

/* SLAM.cpp:553-574 filterDepthImage on the device: every value outside [min_d, max_d] -> 0
 * (:558-566; SLAM.cpp:229 passes 25000 / 1000, SLAM.hpp:15-16), then, if morph != 0, cv::dilate and
 * cv::erode with the 5x5 rectangle of :568-573 in one LDS-tiled pass.  anchor_x / anchor_y: the
 * anchor of dilate / erode inside the element, -1 = its centre (2, 2) -- what OpenCV uses when,
 * as at :572-573, no anchor is passed (the Point(3,3) handed to getStructuringElement only shapes
 * MORPH_CROSS elements).  OpenCV is third-party and absent here: anchor and border rule
 * (out-of-image pixels never win) are PARITY UNPINNED, restated from its documentation.
 * depth_in / depth_out: host arrays of rows x cols (may be the same array). */
int icpk_filter_depth_image(icpk_ctx *ctx, const uint16_t *depth_in, uint16_t *depth_out, int32_t rows,
                            int32_t cols, int32_t max_d, int32_t min_d, int32_t morph, int32_t anchor_x,
                            int32_t anchor_y);
/* filterDepthImage + back-projection without the image leaving the device (SLAM.cpp:229 then
 * icp.cpp:38-39): as icpk_backproject (normals_mode < 0) or icpk_backproject_with_normals
 * (normals_mode >= 0, which must be 1) on the filtered image */
int icpk_backproject_filtered(icpk_ctx *ctx, const uint16_t *depth, int32_t rows, int32_t cols, float fx,
                              float cx, const float offset[3], int32_t which, int32_t normals_mode,
                              int32_t max_d, int32_t min_d, int32_t morph, int32_t anchor_x, int32_t anchor_y);

/* The cloud set-up of icp::getTransformation for one frame pair in ONE call (icp.cpp:38-39 back-project
 * `data` and `previous`, :58-59 / :70-71 rotate both by cameraRotation and translate by cameraPosition):
 * depth_source = the current frame, depth_target = the previous one; offset as in icpk_backproject;
 * R, t (both or neither) = the pose applied to every point afterwards, p <- fl32(fl32(R p) + t); filter != 0
 * runs icpk_filter_depth_image's filter (max_d ... anchor_y) on both frames first.  Equivalent, bit for bit,
 * to icpk_backproject[_filtered] x 2 + icpk_transform_target + icpk_transform_source + icpk_commit_source
 * (the posed source is the starting point of the alignment), with 3-5 kernel launches instead of 25
 * and one host wait instead of two.  *n_source / *n_target: the cloud sizes.
 * depth_target == NULL: the previous frame is the one this context received as depth_source in its last
 * icpk_backproject_pair call (SLAM.cpp:305, previous = filtered.clone(): the caller hands the same frame back) --
 * its image, and its filtered copy if the filter settings are unchanged, are still on the device, so only ONE
 * image crosses PCIe.  Same results as passing the frame again.  ICPK_E_NOT_SET if no frame of this size is
 * resident (first call, other rows / cols, or the image buffers were used by another call in between). */
int icpk_backproject_pair(icpk_ctx *ctx, const uint16_t *depth_source, const uint16_t *depth_target,
                          int32_t rows, int32_t cols, float fx, float cx, const float offset[3],
                          const float R[9], const float t[3], int32_t filter, int32_t max_d, int32_t min_d,
                          int32_t morph, int32_t anchor_x, int32_t anchor_y, int32_t *n_source,
                          int32_t *n_target);

/* ---- point-to-plane extension (BASELINE config 3; not in the reference) ---- */
#define ICPK_NORMALS_CROSS 0     /* normalised cross product of back-projected central differences */
#define ICPK_NORMALS_REFERENCE 1 /* SLAM.cpp:421-425 getNormalMap formula, interior pixels          */
/* target cloud AND one normal per point from a depth image (as icpk_backproject
 * with which = 1); pixels without a normal get (0,0,0) and never pair. */
int icpk_backproject_with_normals(icpk_ctx *ctx, const uint16_t *depth, int32_t rows, int32_t cols,
                                  float fx, float cx, const float offset[3], int32_t normals_mode);
/* normals for the current target cloud from host arrays (n must equal the target size) */
int icpk_set_target_normals(icpk_ctx *ctx, const float *nx, const float *ny, const float *nz, int32_t n);
int icpk_get_target_normals(icpk_ctx *ctx, float *nx, float *ny, float *nz);
/* K5: the 28 canonical sums of the linearised point-to-plane step over the
 * associations of the last sweep (see ICPK_NP2L) */
int icpk_reduce_p2l(icpk_ctx *ctx, float max_dist, double *sums, int64_t *count);
/* host solve of the step: 0 ok, ICPK_W_DEGENERATE if not positive definite */
int icpk_solve_point_to_plane(const double sums[28], double R[9], double t[3]);

/* ---- test hook ------------------------------------------------------------ */
/* icp.cpp:606-620 distance(color_point_t, color_point_t) evaluated on the
 * device for n pairs; a and b are host xyz-SoA arrays [3][n].  Lets the parity
 * tests check the float/double/sqrt sequence bit for bit on its own. */
int icpk_pair_distance(icpk_ctx *ctx, const float *a, const float *b, float *out, int32_t n);
/* same for icp.cpp:595-602 distance(cv::Point3f, cv::Point3f): double sqrt, narrowed on
 * return (dead in the reference: its only call site is pointcloud.cpp:246) */
int icpk_pair_distance3(icpk_ctx *ctx, const float *a, const float *b, float *out, int32_t n);

/* ---- small host helpers restated from the reference (no device work) ----- */
float icpk_distance3(const float a[3], const float b[3]);                            /* icp.cpp:595-602      */
void icpk_make_rotation_matrix(float x_deg, float y_deg, float z_deg, float out[9]); /* icp.cpp:640-653      */
void icpk_matrix_to_quaternion(const float m[9], float q_wxyz[4]);                 /* quaternion.cpp:23-79 */
void icpk_quaternion_to_euler(const float q_wxyz[4], float e_deg[3]);              /* SLAM.cpp:613-636     */
/* pointcloud.cpp:60-98: the 3-D points of a frame's key points (what findGlobalKeyPointAssociations, icp.cpp:488,
 * is fed).  kp_xy: n pixel positions (x, y) as cv::KeyPoint::pt holds them; each is rounded to a pixel as the
 * reference's Point2f -> Point2i conversion does (cvRound: to nearest, ties to even), dropped if its depth is 0
 * (:67-70) -- or if it falls outside the image, where the reference reads out of bounds -- and back-projected with
 * the formula of :86-88 (CX and FX for y too).  out_xyz: room for n points (x, y, z interleaved); kept (or NULL):
 * the index into kp_xy of every point written.  Returns the number of points (>= 0) or a negative status.  Host only. */
int icpk_backproject_keypoints(const uint16_t *depth, int32_t rows, int32_t cols, const float *kp_xy, int32_t n,
                               float fx, float cx, float *out_xyz, int32_t *kept);
/* host solve exposed for testing: reference flavour from the float moment,
 * Kabsch flavour from raw sums (n, sum a, sum b, sum a b^T) */
void icpk_solve_reference(const float M[9], float R[9]);                           /* icp.cpp:215-223      */
void icpk_solve_kabsch(int64_t n, const double sa[3], const double sb[3],
                       const double sab[9], double R[9], double t[3]);             /* rigid_transform_3D.py:9-40 */

#ifdef __cplusplus
}
#endif
#endif /* ICPK_H */
