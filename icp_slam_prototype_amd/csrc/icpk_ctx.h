// icpk_ctx.h -- the context object behind the opaque icpk_ctx of include/icpk.h, shared by
// the host-side translation units (icpk_api.cpp, icpk_comm.cpp).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <string>
#include <vector>

#include "icpk.h"
#include "icpk_internal.h"

namespace icpk {

struct Cloud {
  float* base = nullptr;
  int n = 0;
  int cap = 0;  // floats per plane, multiple of NN_TILE
  float* x() const { return base; }
  float* y() const { return base + cap; }
  float* z() const { return base + 2 * (size_t)cap; }
};

inline int round_up(int v, int m) { return ((v + m - 1) / m) * m; }

}  // namespace icpk

using icpk::nn_key_t;
using icpk::LoopState;
using icpk::GridInfo;
using icpk::NSUM;

struct icpk_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  icpk::Cloud tgt, src0, src;
  icpk::Cloud nrm;  // target normals (point-to-plane), same indexing as tgt
  bool have_normals = false;
  icpk::Cloud dec;  // every NN_SEED_STRIDE-th target (seeding pre-pass of the filtered NN)
  bool have_tgt = false, have_src = false, have_assoc = false;
  bool have_dec = false;   // dec matches tgt
  bool have_boxes = false; // boxes match tgt
  float* boxes = nullptr;  // [6][tbox_stride] tile boxes then [6][sbox_stride] sub-tile boxes
  int boxes_tiles_cap = 0;
  // pruned scan: Morton-ordered copy of the target, its permutation, the query order
  icpk::Cloud sorted;
  int* tperm = nullptr;
  unsigned* tkeys = nullptr;  // sorted Morton codes of the target (first-sweep seeding)
  int tperm_cap = 0;
  int* qperm = nullptr;
  int qperm_cap = 0;
  bool have_qperm = false;
  float* bounds = nullptr;  // 6 floats: lo xyz, hi xyz of the target
  unsigned* sort_keys = nullptr;  // 2 x sort_cap
  int* sort_vals = nullptr;       // sort_cap
  int sort_cap = 0;
  icpk::GridInfo* morton_table = nullptr;  // device: the table size of the Morton counting sort (pruned scan), for launch_grid_scan
  bool have_seed = false;  // `best` holds matches of a previous sweep of the same clouds
  nn_key_t* best = nullptr;
  nn_key_t* seed = nullptr;
  nn_key_t* best_m = nullptr;  // pruned scan: results / seeds in query Morton order
  nn_key_t* seed_m = nullptr;
  bool have_seed_m = false;    // best_m holds the matches of the previous sweep under the current qperm
  int32_t* idx = nullptr;
  float* dist = nullptr;
  int assoc_cap = 0;
  double* partial = nullptr;
  int* pcount = nullptr;
  double* red_out = nullptr;   // device, 20 x 8 bytes
  double* red_host = nullptr;  // pinned, 20 x 8 bytes
  LoopState* st_dev = nullptr;   // device-side loop state
  LoopState* st_host = nullptr;  // pinned staging copy
  int* progress = nullptr;       // pinned, mapped: LoopState::progress of a throttled loop (see there)
  int* progress_dev = nullptr;   // the same words as the device addresses them
  int loop_epoch = 0;            // tag of the current throttled loop in the progress words
  LoopState* st_mirror = nullptr;      // pinned + mapped: the loop's outputs as the device writes them at its end (LoopState::mirror)
  LoopState* st_mirror_dev = nullptr;  // the same memory as the device addresses it
  bool result_mirror = true;           // ICPK_RESULT_MIRROR=0: copy the state back and wait for the stream instead (diagnostic)
  icpk::LoopInitArgs pending_init{};         // device_loop_begin(defer): the initial LoopState not launched yet
  bool init_pending = false;
  int* grid_ticket = nullptr;          // grid_begin_kernel's arrival counter (zero between launches)
  int sub_factor = 0;                  // icpk_set_subsample: keep one valid pixel in sub_factor (<= 1: all), chosen by ...
  unsigned long long sub_seed = 0;     // ... a hash of this seed, the image's stream number and the pixel
  unsigned long long sub_stream = 0;   // images back-projected since icpk_set_subsample (every image draws a fresh pattern, as rand() would)
  struct HostRange {                   // icpk_register_host_buffer: caller memory pinned and mapped for the device
    const char* host;
    size_t bytes;
    const char* dev;
  };
  std::vector<HostRange> registered;
  bool zero_copy_upload = true;        // ICPK_ZERO_COPY_UPLOAD=0: copy-engine transfer from the staging buffer instead (diagnostic)
  uint16_t* stage_depth = nullptr;     // pinned: icpk_backproject_pair's images on their way to the device
  int stage_depth_cap = 0;
  int* pix_tidx = nullptr;             // icpk_backproject_pair: the target point of every pixel (-1: none) ...
  int* pix_src = nullptr;              // ... and the pixel of every source point: image-space seeds of the alignment that follows
  int pix_cap = 0;
  bool have_pix_seed = false;          // they describe the clouds the context holds now
  bool image_order = true;             // ICPK_IMAGE_ORDER=0: sort the queries of an image-ordered source by cell like any other (diagnostic)
  bool lazy_unpack = true;             // ICPK_LAZY_UNPACK=0: unpack the records at the end of every device loop (diagnostic)
  bool pixel_seeds = true;             // ICPK_PIXEL_SEEDS=0: the reference's literal seed (diagnostic)
  int pix_rows = 0, pix_cols = 0;
  bool src_pristine = false;     // the working source equals the committed one (see copy_src0_to_src)
  bool pristine_skip = true;     // ICPK_PRISTINE_SKIP=0: always copy (diagnostic)
  int loop_ahead = 1;            // iterations kept enqueued ahead of the device in a loop that may exit early
  const int* stop = nullptr;     // &st_dev->done while a device loop is being enqueued, else null
  LoopState* st_active = nullptr;  // st_dev while a device loop is being enqueued, else null
  float* stage_t = nullptr;  // pinned staging of host clouds (frame-batch slots): target, source
  float* stage_s = nullptr;
  int stage_t_cap = 0, stage_s_cap = 0;
  uint16_t* depth_dev = nullptr;
  uint16_t* depth_flt = nullptr;  // filtered depth image (icpk_filter_depth_image / icpk_backproject_filtered)
  int32_t* ks_buf = nullptr;      // key-point association lists: assoc_q | assoc_t | assoc_d | rej_q, ks_cap entries each
  int ks_cap = 0;
  int depth_cap = 0;
  // icpk_backproject_pair keeps the current frame's image on the device (slot frame_slot of depth_dev / depth_flt, two
  // slots of depth_cap / 2 pixels): it is the next call's `previous` (SLAM.cpp:305) and need not cross PCIe again
  int frame_slot = -1;               // -1: no resident frame
  int frame_rows = 0, frame_cols = 0;
  int frame_filter[6] = {0, 0, 0, 0, 0, 0};  // filter settings the resident filtered copy was made with (on, max, min, morph, ax, ay)
  int* bp_counts = nullptr;
  int bp_counts_cap = 0;
  int* bp_n_host = nullptr;  // pinned
  std::vector<hipEvent_t> events;
  std::vector<float> trace_R, trace_t, trace_mse;  // per-iteration record of the last align
  std::vector<int32_t> trace_pairs;
  int target_blocks = 16384;
  int q_per_lane = 0;  // 0 = auto
  int slices = 0;      // pruned scan: lanes per query (0 = by cloud size)
  // grid scan (ICPK_NN_GRID): cell table + AoS copy of the target sorted by cell
  icpk::GridInfo* grid_info = nullptr;
  float* grid_bounds = nullptr;
  int* cell_start = nullptr;  // grid_max_cells + 1
  int grid_max_cells = icpk::GRID_MAX_CELLS;  // capacity of cell_start / qcount / qstart (a frame-batch slot: GRID_MAX_CELLS_SLOT)
  float4* t4 = nullptr;
  float4* o4 = nullptr;      // the target in the CALLER's order as (x, y, z, 0): K2's gather of the matched point is one 16-byte load (valid while have_grid)
  float4* qm4 = nullptr;     // queries in scan order (x, y, z, original index)
  float4* sp_in = nullptr;   // seeds as points, scan order: read by the next grid sweep
  float4* sp_out = nullptr;  // ... written by it
  float4* rec = nullptr;     // device loop behind grid sweeps: caller-order records {(query, distance), (match, index)}, 2 float4 per query -- what K2 reads
  bool rec_pending = false;  // the last device loop left planes / keys to be unpacked from qm4 / rec on demand (ensure_unpacked)
  int qm4_cap = 0;
  // frame-batch mode: child contexts (one per pair in flight; own stream for set-up work) --
  // owned by the parent, never handed out
  std::vector<icpk_ctx*> slots;
  hipEvent_t ready_ev = nullptr;       // slot: set-up of the current pair is enqueued up to here
  hipEvent_t group_ev[2] = {nullptr, nullptr};  // parent: the lock-step loop of a slot set has finished
  hipStream_t setup_stream[2] = {nullptr, nullptr};  // parent: the batched set-up launches of a slot set (created on first use)
  hipEvent_t setup_ev[2] = {nullptr, nullptr};       // ... and their completion
  hipEvent_t batch_t0[2] = {nullptr, nullptr}, batch_t1[2] = {nullptr, nullptr};  // parent: params.profile = 1 in the
  bool batch_timed[2] = {false, false};                                           //   frame-batch mode: one sweep per group
  LoopState* slot_states = nullptr;       // parent: the loop states of all slots in one allocation (slot k at [k]),
  LoopState* slot_states_host = nullptr;  //   so that a group's states come back with ONE copy; pinned mirror
  bool st_pooled = false;                 // slot: st_dev / st_host point into the parent's pools
  int batch_setup = 1;                               // 0 (ICPK_BATCH_SETUP=0): per-pair set-up launches on the slots' own streams
  int batch_group = 16;                // pairs advancing in lock step (ICPK_BATCH_GROUP, <= BATCH_MAX)
  int batch_threads = 4;               // host threads sharing a group's set-up calls (ICPK_BATCH_THREADS)
  std::vector<nn_key_t*> best_of_sweep;  // device loop: which buffer each enqueued sweep wrote
  // RCCL communicator of the frame-batch / query-sharded modes (icpk_comm.cpp); null until
  // icpk_comm_init_rccl
  struct icpk_comm_state* comm = nullptr;
  int* qcount = nullptr;     // query counting sort by cell: counts and starts, grid_max_cells + 1 each
  int* qstart = nullptr;
  bool qcount_dirty = false; // a counting sort was cut short: clear the whole count table before the next one
  int* scan_bsum = nullptr;  // block sums of the cell-count scans (GRID_SCAN_BLOCKS ints)
  // a fresh pair sorts targets and queries side by side (build_grid_and_order): second count table, block sums and slots
  int* qcount2 = nullptr;
  bool qcount2_dirty = false;
  int* scan_bsum2 = nullptr;
  int* sort_vals2 = nullptr;
  int sort_vals2_cap = 0;
  int loop_nact = icpk::NSUM;      // device loop: sums the running alignment's step consumes (NSUM_REF or NSUM)
  int profile_phase = 0;     // alignments profiled so far (offsets the sampled launches, see profile_stride)
  int qperm_kind = 0;        // what qperm holds: 1 Morton order (pruned scan), 2 cell order (grid scan)
  bool grid_chain = false;   // device loop only: the previous sweep was a grid sweep (qm4 / sp_in current)
  int t4_cap = 0;
  bool have_grid = false;  // grid matches tgt
  int grid_xdiv = 4;       // cells are this many times finer along x (ICPK_GRID_XDIV; measured best on config 2: 4)
  float grid_ppc = 8.f;    // aimed-at targets per occupied cell (ICPK_GRID_PPC; measured best on configs 2, 3 and the dense pair: 8)
  int grid_slices = 0;     // lanes per query (0 = by cloud size)
  int merged_setup = 1;    // a fresh pair's two counting sorts side by side (ICPK_MERGED_SETUP=0: one after the other)
  std::string err;
  icpk_log_fn log_fn = nullptr;
  void* log_user = nullptr;
  std::chrono::steady_clock::time_point log_last;
};

#define ICPK_HIP(ctx, call)                                                                      \
  do {                                                                                           \
    hipError_t e__ = (call);                                                                     \
    if (e__ != hipSuccess) {                                                                     \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                           \
      return ICPK_E_HIP;                                                                         \
    }                                                                                            \
  } while (0)

// helpers of icpk_api.cpp that icpk_comm.cpp needs
int icpk_host_fail(icpk_ctx* ctx, int code, const char* msg);
int icpk_host_ensure_cloud(icpk_ctx* ctx, icpk::Cloud& c, int n);
// the target cloud of ctx has been replaced on the device (n points, planes filled up to n):
// pad it and drop everything derived from the previous target
int icpk_host_target_replaced(icpk_ctx* ctx);
void icpk_comm_release(icpk_ctx* ctx);  // called by icpk_destroy
int icpk_comm_allreduce_device(icpk_ctx* ctx, double* dev, int n);  // in-stream sum over the ranks (icpk_comm.cpp)
