// kernels_grid.hip -- K1d: exact NN through a uniform grid over the target (ICPK_NN_GRID).
//
// SURVEY.md 8(f) rank 2 again (the reference's own plan is a voxel look-up, icp.cpp:347-486,
// map.hpp:9-17), this time with the voxels as the index itself: targets are binned into
// cubic cells of edge h, sorted by linear cell id (x fastest), and `cell_start` gives the
// first sorted position of every cell, so the targets of any run of x-adjacent cells are one
// contiguous range.  A query with a seed match at distance r only has to look at the cells
// that meet the cube [q - r, q + r]: (cells in y) x (cells in z) contiguous ranges, a few
// dozen to a few hundred targets instead of all Nt -- and still returns the brute-force
// result bit for bit: every candidate goes through the same fp32 filter + float64 exact
// re-evaluation + lexicographic (distance, index) merge as the other kernels, and the cell
// cube is a superset of every target that could tie or beat the seed (see cube_cells).
//
// Set-up (once per target cloud): finite bounds, cell size from the point density, counting
// sort by cell (slot by atomics + a device-sized exclusive scan = the cell starts) into an AoS copy
// (x, y, z, original index: one 16-byte load per candidate).
#include <type_traits>

#include "icpk_internal.h"
#include "loop_init.h"
#include "nn_device.h"
#include "wave_sum.h"

namespace icpk {

// ---- set-up ---------------------------------------------------------------------------
// bounds over the FINITE coordinates only (a non-finite target can never be selected: its
// distance is inf/NaN, and the seed of such a query is element 0, the lowest index)
__device__ __forceinline__ void grid_bounds_body(const BoundsArgs& a, const int part) {
  __shared__ float red[6][16];
  float lo[3], hi[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    lo[c] = __builtin_inff();
    hi[c] = -__builtin_inff();
  }
  // (the three planes in one loop: their loads are in flight together -- one memory phase, not three)
  for (int i = part * 1024 + threadIdx.x; i < a.n; i += a.nparts * 1024) {
    const float v[3] = {a.x[i], a.y[i], a.z[i]};
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (v[c] - v[c] == 0.f) {
        lo[c] = __builtin_fminf(lo[c], v[c]);
        hi[c] = __builtin_fmaxf(hi[c], v[c]);
      }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      lo[c] = __builtin_fminf(lo[c], __shfl_xor(lo[c], m, 64));
      hi[c] = __builtin_fmaxf(hi[c], __shfl_xor(hi[c], m, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      red[c][threadIdx.x >> 6] = lo[c];
      red[3 + c][threadIdx.x >> 6] = hi[c];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = red[threadIdx.x][0];
    for (int k = 1; k < 16; ++k)
      v = threadIdx.x < 3 ? __builtin_fminf(v, red[threadIdx.x][k]) : __builtin_fmaxf(v, red[threadIdx.x][k]);
    a.fb[part * 6 + threadIdx.x] = v;  // one partial box per block; grid_info_kernel merges them
  }
}
__global__ __launch_bounds__(1024) void grid_bounds_kernel(const BoundsArgs a) { grid_bounds_body(a, blockIdx.x); }
// frame-batch set-up kernels: blockIdx.y = pair of the group; workgroups beyond a pair's own count leave at once
__global__ __launch_bounds__(1024) void grid_bounds_batch_kernel(const SetupBatchOf<BoundsArgs> b) {
  const BoundsArgs& a = b.p[blockIdx.y];
  if ((int)blockIdx.x < a.nparts) grid_bounds_body(a, blockIdx.x);
}

// cell edge: about `ppc` targets per occupied cell if the cloud is a surface whose area is
// of the order of the bounding box's faces (a depth image is); never more than
// the context's table capacity (GRID_MAX_CELLS; GRID_MAX_CELLS_SLOT for a frame-batch slot).  Only efficiency depends on the choice.
// Cells are `xdiv` times finer along x, the axis the sorted order runs along: a query's cube
// costs one contiguous range per (y, z) row whatever the x resolution, so finer x cells trim
// the ranges to the cube (fewer candidates outside it) at no extra look-up.
__device__ __forceinline__ void grid_info_body(const float* __restrict__ fbp, int nparts, int n, float ppc, int xdiv,
                                               int max_cells, GridInfo* __restrict__ g) {
  // one wave: lane l merges the partial boxes l, l + 64, ...; xor butterfly; lane 0 goes on
  float fb[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    float v = c < 3 ? __builtin_inff() : -__builtin_inff();
    for (int b = threadIdx.x; b < nparts; b += 64)
      v = c < 3 ? __builtin_fminf(v, fbp[b * 6 + c]) : __builtin_fmaxf(v, fbp[b * 6 + c]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const float o = __shfl_xor(v, m, 64);
      v = c < 3 ? __builtin_fminf(v, o) : __builtin_fmaxf(v, o);
    }
    fb[c] = v;
  }
  if (threadIdx.x != 0) return;
  float lo[3], ext[3];
  float emax = 0.f;
  for (int c = 0; c < 3; ++c) {
    const bool ok = fb[c] <= fb[3 + c];  // false if the cloud has no finite point
    lo[c] = ok ? fb[c] : 0.f;
    ext[c] = ok ? fb[3 + c] - fb[c] : 0.f;
    if (!(ext[c] >= 0.f) || !(ext[c] <= 3.0e38f)) ext[c] = 0.f;
    emax = __builtin_fmaxf(emax, ext[c]);
  }
  const float area = ext[0] * ext[1] + ext[1] * ext[2] + ext[2] * ext[0];
  float h = __builtin_sqrtf(ppc * area / (float)(n > 0 ? n : 1));
  if (!(h > 0.f) || !(h <= 3.0e38f)) h = emax > 0.f ? emax / 64.f : 1.f;  // a line or a single point
  h = __builtin_fmaxf(h, emax / 1023.f);  // <= 1024 cells per axis
  h = __builtin_fmaxf(h, 1e-30f);
  int nx, ny, nz;
  float hx;
  for (;;) {
    hx = h / (float)xdiv;
    nx = (int)(ext[0] / hx) + 1;
    ny = (int)(ext[1] / h) + 1;
    nz = (int)(ext[2] / h) + 1;
    if ((long long)nx * ny * nz <= max_cells) break;
#ifndef ICPK_GRID_GROW
#define ICPK_GRID_GROW 1.06f  // (1.26 left up to half of the table unused: a cell edge 20 % longer than necessary)
#endif
    h *= ICPK_GRID_GROW;
  }
  g->lo[0] = lo[0];
  g->lo[1] = lo[1];
  g->lo[2] = lo[2];
  g->h = h;
  g->inv_h = 1.0f / h;
  g->inv_hx = 1.0f / hx;
  g->xdiv = xdiv;
  g->nx = nx;
  g->ny = ny;
  g->nz = nz;
  g->ncells = nx * ny * nz;
  // the query order (locality only) uses cells xdiv times coarser along x: a shorter count table
  g->nxq = (nx + xdiv - 1) / xdiv;
  g->ncells_q = g->nxq * ny * nz;
}

__global__ void grid_info_kernel(const InfoArgs a) { grid_info_body(a.fb, a.nparts, a.n, a.ppc, a.xdiv, a.max_cells, a.g); }

// The first launch of a fresh frame pair's set-up (build_grid_and_order): the bounds pass, and behind it -- in the
// workgroup that draws the last ticket -- the grid's geometry; workgroup 0 writes the initial LoopState of the
// alignment on the side (has_init).  One launch instead of three on a path that is a chain of tiny dependent
// kernels, each of which costs ~4 us of dispatch on top of its ~3 us of work.  The hand-off is the placement-
// independent one: partial boxes stored, agent-scope release, ticket; the last arriver acquires and reads them.
__global__ __launch_bounds__(1024) void grid_begin_kernel(const BoundsArgs a, const InfoArgs info, const LoopInitArgs li,
                                                          const int has_init, int* __restrict__ ticket) {
  if (has_init && blockIdx.x == 0) loop_init_body(li);
  grid_bounds_body(a, blockIdx.x);
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == a.nparts - 1;
    if (s_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (ready for the next launch)
    }
  }
  __syncthreads();
  if (!s_last || threadIdx.x >= 64) return;
  grid_info_body(info.fb, info.nparts, info.n, info.ppc, info.xdiv, info.max_cells, info.g);
}
__global__ void grid_info_batch_kernel(const SetupBatchOf<InfoArgs> b) {
  const InfoArgs& a = b.p[blockIdx.x];
  grid_info_body(a.fb, a.nparts, a.n, a.ppc, a.xdiv, a.max_cells, a.g);
}

// The ONE mapping coordinate -> cell index along an axis, used for targets and for the
// corners of a query's cube alike.  Every step is monotone non-decreasing in v (float
// subtraction, multiplication by a positive constant, clamp, truncation of a non-negative
// value), so a <= t <= b implies cell(a) <= cell(t) <= cell(b).  NaN maps to cell 0.
__device__ __forceinline__ int grid_cell(float v, float lo, float inv_h, int n) {
  float f = (v - lo) * inv_h;
  f = __builtin_fminf(__builtin_fmaxf(f, 0.f), (float)(n - 1));
  return (int)f;
}

// Counting sort by cell, used for the targets (set-up) and for the query order (once per
// alignment): slot within the cell by atomics -- the order inside a cell may be whatever the
// atomics make it: target candidates are merged lexicographically, and the query order only
// matters for locality (results are scattered back by original index).
__device__ __forceinline__ void grid_qslot_body(const QslotArgs& a, const int block) {
  const int i = block * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const GridInfo g = *a.gi;
  const int cx = grid_cell(a.x[i], g.lo[0], g.inv_hx, g.nx);
  const int cy = grid_cell(a.y[i], g.lo[1], g.inv_h, g.ny);
  const int cz = grid_cell(a.z[i], g.lo[2], g.inv_h, g.nz);
  const int c = a.coarse ? (cz * g.ny + cy) * g.nxq + cx / g.xdiv : (cz * g.ny + cy) * g.nx + cx;
  a.cell[i] = c;
  a.slot[i] = atomicAdd(&a.count[c], 1);
}
__global__ void grid_qslot_kernel(const QslotArgs a) { grid_qslot_body(a, blockIdx.x); }
__global__ void grid_qslot_batch_kernel(const SetupBatchOf<QslotArgs> b) { grid_qslot_body(b.p[blockIdx.y], blockIdx.x); }

// qm4 != nullptr: the first sweep of an alignment without seeds -- the queries in scan order (x, y, z, index),
// the reference's literal seed (element 0, icp.cpp:572) as a point and as a key, written here instead of by
// a zero fill and grid_query_points_kernel afterwards (two launches less per alignment)
__device__ __forceinline__ void grid_qscatter_body(const QscatterArgs& a, const int block) {
  const int i = block * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const int ip = a.qcell ? a.qstart[a.qcell[i]] + a.qslot[i] : i;  // (qcell == nullptr: the caller's order is the scan order)
  a.qperm[ip] = i;
  if (a.qm4) {
    a.qm4[ip] = make_float4(a.qx[i], a.qy[i], a.qz[i], __int_as_float(i));
    int j = 0;  // the reference's literal seed
    if (a.spix) {  // the target point of the query's own pixel, else of the nearest occupied pixel of the 5 x 5 around it
      const int p = a.spix[i];
      const int r = p / a.cols, c = p - r * a.cols;
      int found = a.tidx[p];
      // (a ring's pixels are read together -- clamped addresses, no branch around a load -- and the nearest occupied one wins)
      auto ring_search = [&](auto ring_c) {
        constexpr int RING = decltype(ring_c)::value;
        int best = -1, best_d = 1 << 30;
#pragma unroll
        for (int dr = -RING; dr <= RING; ++dr) {
#pragma unroll
          for (int dc = -RING; dc <= RING; ++dc) {
            if ((dr > -RING && dr < RING) && (dc > -RING && dc < RING)) continue;  // (the inner rings have been looked at)
            const int rr = r + dr, cc = c + dc;
            const bool inside = rr >= 0 && rr < a.rows && cc >= 0 && cc < a.cols;
            const int t = a.tidx[min(max(rr, 0), a.rows - 1) * a.cols + min(max(cc, 0), a.cols - 1)];
            const int d2 = dr * dr + dc * dc;
            if (inside && t >= 0 && d2 < best_d) {
              best_d = d2;
              best = t;
            }
          }
        }
        return best;
      };
      if (found < 0) found = ring_search(std::integral_constant<int, 1>{});
      if (found < 0) found = ring_search(std::integral_constant<int, 2>{});
      if (found >= 0) j = found;
    }
    a.sp[ip] = make_float4(a.ox[j], a.oy[j], a.oz[j], __int_as_float(j));
    a.seed_m[ip] = 0ull;
  }
}
__global__ void grid_qscatter_kernel(const QscatterArgs a) { grid_qscatter_body(a, blockIdx.x); }
__global__ void grid_qscatter_batch_kernel(const SetupBatchOf<QscatterArgs> b) { grid_qscatter_body(b.p[blockIdx.y], blockIdx.x); }

// ---- device-sized exclusive scan of the cell counts ----------------------------------------
// The number of cells lives in GridInfo ON THE DEVICE; these kernels read it there, so the whole
// grid build is enqueued without a host round trip (the frame-batch mode builds the grids of the
// next group while the current group's loop keeps the GPU busy: a host wait there costs
// milliseconds).  A fixed launch geometry walks whatever the table size turns out to be.  Scan in two
// launches: 2048 counts per tile (256 lanes x 8) -> tile sums; then every tile adds up the sums of the tiles
// before it (<= 4097 values, 16 per lane) and scans its own counts from there -- and leaves them ZERO: the
// count table is all zero between two sorts
// (hipMemset at allocation), so no sort starts with a zero-fill launch.
constexpr int GSCAN_ITEMS = 8;
constexpr int GSCAN_TILE = 256 * GSCAN_ITEMS;
static_assert(GSCAN_ITEMS == 8, "grid_scan_apply_body unpacks two int4");
// (the largest table, 2^23 + 1 entries, has 4097 tiles)

__device__ __forceinline__ int grid_table_size(const GridInfo* __restrict__ gi, int coarse) {
  return (coarse ? gi->ncells_q : gi->ncells) + 1;
}

__device__ __forceinline__ int block_sum_256(int v, int* sh) {  // sum over the 256 lanes, in every lane
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const int t = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return t;
}

// Both kernels walk the table's tiles with the stride of the launch: GSCAN_LAUNCH_BLOCKS workgroups serve any table
// size (a Kinect-size grid has ~430 tiles; launching the 4097 workgroups of the largest table cost ~3 us per kernel in
// workgroups that left at once).
constexpr int GSCAN_LAUNCH_BLOCKS = 512;

__device__ __forceinline__ void grid_scan_sums_body(const ScanArgs& a, const int block, const int nblocks) {
  __shared__ int sh[4];
  const int n = grid_table_size(a.g, a.coarse);
  for (int tile = block; tile * GSCAN_TILE < n; tile += nblocks) {
    const int base = tile * GSCAN_TILE;
    int v = 0;
#pragma unroll
    for (int k = 0; k < GSCAN_ITEMS; ++k) {
      const int i = base + k * 256 + threadIdx.x;
      v += i < n ? a.count[i] : 0;
    }
    const int t = block_sum_256(v, sh);
    if (threadIdx.x == 0) a.bsum[tile] = t;
  }
}
__global__ __launch_bounds__(256) void grid_scan_sums_kernel(const ScanArgs a) { grid_scan_sums_body(a, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(256) void grid_scan_sums_batch_kernel(const SetupBatchOf<ScanArgs> b) {
  grid_scan_sums_body(b.p[blockIdx.y], blockIdx.x, gridDim.x);
}

__device__ __forceinline__ void grid_scan_apply_body(const ScanArgs& a, const int block, const int nblocks) {
  __shared__ int wtot[4];
  __shared__ int sh[4];
  int* __restrict__ in = a.count;
  const int n = grid_table_size(a.g, a.coarse);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int tile = block; tile * GSCAN_TILE < n; tile += nblocks) {
    const int base = tile * GSCAN_TILE;
    // this tile's offset: the sums of the tiles before it
    int before = 0;
    for (int b = t; b < tile; b += 256) before += a.bsum[b];
    const int boff = block_sum_256(before, sh);
    // lane t owns the 8 consecutive counts base + 8 t .. base + 8 t + 7
    int c[GSCAN_ITEMS], s = 0;
    const bool whole = base + GSCAN_TILE <= n;  // (all tiles but the last: 16-byte loads and stores, the tables are 16-byte aligned)
    if (whole) {
      const int4 lo4 = *reinterpret_cast<const int4*>(in + base + GSCAN_ITEMS * t);
      const int4 hi4 = *reinterpret_cast<const int4*>(in + base + GSCAN_ITEMS * t + 4);
      c[0] = lo4.x, c[1] = lo4.y, c[2] = lo4.z, c[3] = lo4.w, c[4] = hi4.x, c[5] = hi4.y, c[6] = hi4.z, c[7] = hi4.w;
#pragma unroll
      for (int k = 0; k < GSCAN_ITEMS; ++k) s += c[k];
    } else {
#pragma unroll
      for (int k = 0; k < GSCAN_ITEMS; ++k) {
        const int i = base + GSCAN_ITEMS * t + k;
        c[k] = i < n ? in[i] : 0;
        s += c[k];
      }
    }
    int inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    __syncthreads();  // (wtot of the previous tile has been read)
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int run = boff + inc - s;
    for (int w = 0; w < wave; ++w) run += wtot[w];
    if (whole) {
      int o[GSCAN_ITEMS];
#pragma unroll
      for (int k = 0; k < GSCAN_ITEMS; ++k) {
        o[k] = run;
        run += c[k];
      }
      int* const op = a.out + base + GSCAN_ITEMS * t;
      *reinterpret_cast<int4*>(op) = make_int4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<int4*>(op + 4) = make_int4(o[4], o[5], o[6], o[7]);
      *reinterpret_cast<int4*>(in + base + GSCAN_ITEMS * t) = make_int4(0, 0, 0, 0);  // the table is handed back all zero
      *reinterpret_cast<int4*>(in + base + GSCAN_ITEMS * t + 4) = make_int4(0, 0, 0, 0);
    } else {
#pragma unroll
      for (int k = 0; k < GSCAN_ITEMS; ++k) {
        const int i = base + GSCAN_ITEMS * t + k;
        if (i < n) {
          a.out[i] = run;
          in[i] = 0;
        }
        run += c[k];
      }
    }
  }
}
__global__ __launch_bounds__(256) void grid_scan_apply_kernel(const ScanArgs a) { grid_scan_apply_body(a, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(256) void grid_scan_apply_batch_kernel(const SetupBatchOf<ScanArgs> b) {
  grid_scan_apply_body(b.p[blockIdx.y], blockIdx.x, gridDim.x);
}

// coarse = 1: the table of the query order (ncells_q entries), else the targets' (ncells).
// out[i] = sum of count[0 .. i) for i in [0, size]; count[] is zero afterwards; bsum: GRID_SCAN_BLOCKS ints
#define ICPK_RECORD(KIND, FIELD, VALUE)           \
  if (SetupRecorder* r__ = setup_recorder()) {    \
    if (r__->n < SETUP_MAX_CALLS) {                            \
      r__->calls[r__->n].kind = KIND;             \
      r__->calls[r__->n++].FIELD = VALUE;         \
    } else {                                      \
      r__->overflow = true;                       \
    }                                             \
    return;                                       \
  }

void launch_grid_scan(int* count, int* out, int* bsum, const GridInfo* g, int coarse, hipStream_t s) {
  const ScanArgs a{count, out, bsum, g, coarse, 0};
  ICPK_RECORD(SK_SCAN, scan, a)
  hipLaunchKernelGGL(grid_scan_sums_kernel, dim3(GSCAN_LAUNCH_BLOCKS), dim3(256), 0, s, a);
  hipLaunchKernelGGL(grid_scan_apply_kernel, dim3(GSCAN_LAUNCH_BLOCKS), dim3(256), 0, s, a);
}
void launch_grid_scan_batch(const SetupBatchOf<ScanArgs>& b, int count, hipStream_t s) {
  if (count <= 0) return;
  hipLaunchKernelGGL(grid_scan_sums_batch_kernel, dim3(GSCAN_LAUNCH_BLOCKS, count), dim3(256), 0, s, b);
  hipLaunchKernelGGL(grid_scan_apply_batch_kernel, dim3(GSCAN_LAUNCH_BLOCKS, count), dim3(256), 0, s, b);
}

// targets into the AoS copy (x, y, z, original index), one 16-byte load per candidate -- and, in the
// caller's order, into o4 (x, y, z, 0): K2 fetches a matched point with one load instead of three
__device__ __forceinline__ void grid_tscatter_body(const TscatterArgs& a, const int block) {
  const int i = block * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const float px = a.x[i], py = a.y[i], pz = a.z[i];
  a.t4[a.cell_start[a.tcell[i]] + a.tslot[i]] = make_float4(px, py, pz, __int_as_float(i));
  a.o4[i] = make_float4(px, py, pz, 0.f);
}
__global__ void grid_tscatter_kernel(const TscatterArgs a) { grid_tscatter_body(a, blockIdx.x); }
__global__ void grid_tscatter_batch_kernel(const SetupBatchOf<TscatterArgs> b) { grid_tscatter_body(b.p[blockIdx.y], blockIdx.x); }

template <typename A>
static int max_n(const SetupBatchOf<A>& b, int count) {
  int m = 0;
  for (int k = 0; k < count; ++k) m = b.p[k].n > m ? b.p[k].n : m;
  return m;
}

void launch_grid_tscatter(const float* x, const float* y, const float* z, const int* tcell, const int* tslot,
                          const int* cell_start, int n, float4* t4, float4* o4, hipStream_t s) {
  if (n <= 0) return;
  const TscatterArgs a{x, y, z, tcell, tslot, cell_start, t4, o4, n, 0};
  ICPK_RECORD(SK_TSCATTER, tscatter, a)
  hipLaunchKernelGGL(grid_tscatter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a);
}
void launch_grid_tscatter_batch(const SetupBatchOf<TscatterArgs>& b, int count, hipStream_t s) {
  const int m = max_n(b, count);
  if (count > 0 && m > 0) hipLaunchKernelGGL(grid_tscatter_batch_kernel, dim3((m + 255) / 256, count), dim3(256), 0, s, b);
}

void launch_grid_qslot(const float* x, const float* y, const float* z, int n, const GridInfo* g, int* count, int* qcell,
                       int* qslot, int coarse, hipStream_t s) {
  if (n <= 0) return;
  const QslotArgs a{x, y, z, g, count, qcell, qslot, n, coarse};
  ICPK_RECORD(SK_QSLOT, qslot, a)
  hipLaunchKernelGGL(grid_qslot_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a);
}
void launch_grid_qslot_batch(const SetupBatchOf<QslotArgs>& b, int count, hipStream_t s) {
  const int m = max_n(b, count);
  if (count > 0 && m > 0) hipLaunchKernelGGL(grid_qslot_batch_kernel, dim3((m + 255) / 256, count), dim3(256), 0, s, b);
}
void launch_grid_qscatter(const int* qcell, const int* qslot, const int* qstart, int n, int* qperm, const float* qx,
                          const float* qy, const float* qz, const float* ox, const float* oy, const float* oz,
                          float4* qm4, float4* sp, nn_key_t* seed_m, hipStream_t s) {
  if (n <= 0) return;
  const QscatterArgs a{qcell, qslot, qstart, qperm, qx, qy, qz, ox, oy, oz, qm4, sp, seed_m, n, 0, nullptr, nullptr, 0, 0};
  ICPK_RECORD(SK_QSCATTER, qscatter, a)
  hipLaunchKernelGGL(grid_qscatter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a);
}
void launch_grid_qscatter_batch(const SetupBatchOf<QscatterArgs>& b, int count, hipStream_t s) {
  const int m = max_n(b, count);
  if (count > 0 && m > 0) hipLaunchKernelGGL(grid_qscatter_batch_kernel, dim3((m + 255) / 256, count), dim3(256), 0, s, b);
}

// the two scatters of a fresh pair's set-up (targets into the cell-sorted copies, queries into scan order) in one launch
__global__ void grid_tqscatter_kernel(const TscatterArgs t, const QscatterArgs q) {
  if (blockIdx.y == 0)
    grid_tscatter_body(t, blockIdx.x);
  else
    grid_qscatter_body(q, blockIdx.x);
}
void launch_grid_tqscatter(const TscatterArgs& t, const QscatterArgs& q, hipStream_t s) {
  const int m = t.n > q.n ? t.n : q.n;
  if (m > 0) hipLaunchKernelGGL(grid_tqscatter_kernel, dim3((m + 255) / 256, 2), dim3(256), 0, s, t, q);
}

int grid_bounds_parts(int n) {  // one 1024-thread block per 2048 points, at most GRID_BOUNDS_PARTS
  int nb = (n + 2047) / 2048;
  return nb < 1 ? 1 : (nb > GRID_BOUNDS_PARTS ? GRID_BOUNDS_PARTS : nb);
}
void launch_grid_bounds(const float* x, const float* y, const float* z, int n, float* fb, hipStream_t s) {
  const BoundsArgs a{x, y, z, fb, n, grid_bounds_parts(n)};
  ICPK_RECORD(SK_BOUNDS, bounds, a)
  hipLaunchKernelGGL(grid_bounds_kernel, dim3(a.nparts), dim3(1024), 0, s, a);
}
void launch_grid_bounds_batch(const SetupBatchOf<BoundsArgs>& b, int count, hipStream_t s) {
  int m = 0;
  for (int k = 0; k < count; ++k) m = b.p[k].nparts > m ? b.p[k].nparts : m;
  if (count > 0 && m > 0) hipLaunchKernelGGL(grid_bounds_batch_kernel, dim3(m, count), dim3(1024), 0, s, b);
}
void launch_grid_begin(const float* x, const float* y, const float* z, int n, float* fb, float ppc, int xdiv, int max_cells,
                       GridInfo* g, const LoopInitArgs* init, int* ticket, hipStream_t s) {
  const BoundsArgs a{x, y, z, fb, n, grid_bounds_parts(n)};
  const InfoArgs info{fb, g, a.nparts, n, ppc, xdiv < 1 ? 1 : xdiv, max_cells < 64 ? 64 : max_cells, 0};
  hipLaunchKernelGGL(grid_begin_kernel, dim3(a.nparts), dim3(1024), 0, s, a, info, init ? *init : LoopInitArgs{}, init ? 1 : 0,
                     ticket);
}
void launch_grid_info(const float* fb, int n, float ppc, int xdiv, int max_cells, GridInfo* g, hipStream_t s) {
  const InfoArgs a{fb, g, grid_bounds_parts(n), n, ppc, xdiv < 1 ? 1 : xdiv, max_cells < 64 ? 64 : max_cells, 0};
  ICPK_RECORD(SK_INFO, info, a)
  hipLaunchKernelGGL(grid_info_kernel, dim3(1), dim3(64), 0, s, a);
}
void launch_grid_info_batch(const SetupBatchOf<InfoArgs>& b, int count, hipStream_t s) {
  if (count > 0) hipLaunchKernelGGL(grid_info_batch_kernel, dim3(count), dim3(64), 0, s, b);
}
#undef ICPK_RECORD

// ---- the sweep --------------------------------------------------------------------------
// Cells met by the cube [q - rr, q + rr], rr slightly above the current best distance r.
// A target t that could tie or beat r has d_t = sqrtf(fl32(S_t)) <= r, S_t the float64 sum of
// the squared float differences (icp.cpp:606-620).  sqrtf is correctly rounded, so
// fl32(S_t) <= r^2 (1 + 2^-22); the narrowing of S_t carries a RELATIVE error 2^-24 when the
// radicand is a normal float but an ABSOLUTE error up to 2^-150 when it is a float denormal,
// hence S_t <= r^2 (1 + 2^-21) + 2^-150 and, per axis,
//   |fl(q_c - t_c)| <= sqrt(S_t) <= r (1 + 2^-22) + 2^-75,
//   |q_c - t_c| <= |fl(q_c - t_c)| (1 + 2^-23)   (exact when the difference is denormal).
// rr = r (1 + 2^-19) + |q_c| 2^-21 + 2^-74 covers both and absorbs the rounding of
// q_c -/+ rr, so fl(q_c - rr) <= t_c <= fl(q_c + rr) and, grid_cell being monotone, the
// target's cell lies in [c0, c1].  (Round 1 used 2^-100 as the absolute term, which is too
// small below a cloud scale of ~1e-19: sqrt(2^-150) = 2^-75.)
// ball_trim.  The cube is the box of a ball: a (y, z) row of cells off the query's own row only matters where
// the ball of the best distance reaches it.  With D = |q - t| in real arithmetic, the chain above gives
//   D^2 = sum (q_c - t_c)^2 <= (1 + 2^-23)^2 sum fl(q_c - t_c)^2 <= (1 + 2^-22)(1 + 2^-51) S_t
//       <= r^2 (1 + 2^-20) + 2^-149,   hence   D <= B := r (1 + 2^-19) + 2^-74   (fp32 rounding of B included).
// A target in cell i of an axis (grid_cell: i <= fl(fl(t - lo) inv_h) < i + 1, unless i is the first or last
// cell, which also take everything clamped from outside) has  lo + i h - e <= t <= lo + (i + 1) h + e  with
// e <= 2^-22 n h, so the gap between q and the cell's interval, evaluated in fp32 (errors <= 2^-23 (|q| + |lo|
// + n h)) and reduced by slop = 2^-20 (|q| + |lo| + n h), is a LOWER bound of |q_c - t_c|; the outer side of
// an edge cell is unbounded.  For a target of row (iy, iz) therefore
//   (q_x - t_x)^2 <= B^2 - gap_y^2 - gap_z^2,
// evaluated with B^2 rounded up and the gaps' squares rounded down: negative -> the row holds nothing that
// matters; else the x cells come from q_x -/+ (sqrt(.) (1 + 2^-19) + |q_x| 2^-21 + 2^-74) as the cube's do (the
// root by the bare v_sqrt_f32, <= 1 ulp low, of an argument clamped to >= 2^-100; only for 2^-40 <= B <= 2^60).  Candidates per query on config 2: 32 -> see DESIGN.md.
// pair_dist of nn_device.h from the differences the fp32 filter already holds, one conversion at a time (the
// same operations, hence the same bits; fewer float64 temporaries alive at the kernel's register peak)
__device__ __forceinline__ float pair_dist_seq(float dx, float dy, float dz) {
  double t = (double)dx;
  double s = t * t;
  __builtin_amdgcn_sched_barrier(0);
  t = (double)dy;
  s = __builtin_fma(t, t, s);
  __builtin_amdgcn_sched_barrier(0);
  t = (double)dz;
  s = __builtin_fma(t, t, s);
  return __builtin_sqrtf((float)s);
}

__device__ __forceinline__ float axis_gap(float q, float lo, float h, int i, int n, float slop) {
  const float a = __builtin_fmaf((float)i, h, lo);  // lower edge of cell i
  const float below = i > 0 ? a - q : 0.f;           // (cell 0 extends to -inf)
  const float above = i < n - 1 ? q - (a + h) : 0.f;  // (cell n - 1 extends to +inf)
  return __builtin_fmaxf(__builtin_fmaxf(below, above) - slop, 0.f);
}

__device__ __forceinline__ void cube_cells(float q, float r, float lo, float inv_h, int n, int& c0, int& c1) {
  const float rr = __builtin_fmaf(r, 1.0f + 0x1p-19f, __builtin_fmaf(__builtin_fabsf(q), 0x1p-21f, 0x1p-74f));
  c0 = grid_cell(q - rr, lo, inv_h, n);
  c1 = grid_cell(q + rr, lo, inv_h, n);
}

// S adjacent lanes per query; queries in cell order (neighbouring queries read the same
// rows), seeds / results in that order too, K3 fused in device-loop mode.
// Diagnostic build only (tools/stamp_grid.py, -DICPK_GRID_STAMPS): per-wave wall_clock64
// (100 MHz) stamps of the phases and work counters.
#ifdef ICPK_GRID_STAMPS
__device__ unsigned long long grid_dbg[8 * 16384];
#ifdef ICPK_GRID_COUNTS  // counters only: the slots hold counts, no time stamps
#define GRID_STAMP(k)
#else
#define GRID_STAMP(k)                                                                         \
  do {                                                                                        \
    if ((threadIdx.x & 63) == 0 && wave_id < 16384) grid_dbg[wave_id * 8 + (k)] = wall_clock64(); \
  } while (0)
#endif
#ifdef ICPK_GRID_COUNTS
#define GRID_COUNT(k, v)                                                                      \
  do {                                                                                        \
    if (wave_id < 16384) atomicAdd(&grid_dbg[wave_id * 8 + (k)], (unsigned long long)(v));    \
  } while (0)
#else
#define GRID_COUNT(k, v)
#endif
extern "C" int icpk_debug_read_grid_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(grid_dbg), sizeof(grid_dbg));
}
extern "C" int icpk_debug_clear_grid_stamps() {
  static unsigned long long zero[8 * 16384];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(grid_dbg), zero, sizeof(zero));
}
#else
#define GRID_STAMP(k)
#define GRID_COUNT(k, v)
#endif

#define ICPK_GRID_BLOCK 64  // one wave per workgroup (the row table in LDS relies on it)
#ifdef ICPK_NO_BALL_TRIM  // (diagnostic builds: the cube alone)
constexpr bool GRID_BALL_TRIM = false;
#else
constexpr bool GRID_BALL_TRIM = true;
#endif
// The sweep of ONE frame pair by workgroup `block` of its launch.  nn_grid_kernel runs it for
// a single pair (arguments by value); nn_grid_batch_kernel runs blockIdx.y-many independent
// pairs in lock step (frame-batch mode): same code, same results.
// REC (the sweeps of a device-side loop): the only caller-order output is rec[i] = {(moved query, distance), (matched
// point, index)}, two adjacent 16-byte stores, which is all K2 reads; the caller's planes and keys (three 4-byte and one
// 8-byte scattered store per query and sweep, each paying for a whole sector) are written ONCE after the loop by
// grid_unpack_kernel.  Same values either way.
__device__ __forceinline__ int bcnt_acc(unsigned x, int acc) {  // popcount(x) + acc in one instruction
  int r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
}

template <int S, bool EXPAND, bool REC>
__device__ __forceinline__ void nn_grid_body(
    float* __restrict__ qxp, float* __restrict__ qyp, float* __restrict__ qzp, const int nq, float4* __restrict__ qm4,
    const float4* __restrict__ t4, const int* __restrict__ cell_start, const GridInfo* __restrict__ gi,
    const float* __restrict__ oxp, const float* __restrict__ oyp, const float* __restrict__ ozp,
    const float4* __restrict__ sp_in, float4* __restrict__ sp_out, nn_key_t* __restrict__ best,
    nn_key_t* __restrict__ best_m, float4* __restrict__ rec, const LoopState* __restrict__ st, const int block) {
  // qm4: the queries in scan order (by grid cell), (x, y, z, original index) -- one coalesced
  // 16-byte load instead of the qperm -> coordinates chain; kept in step with the caller's
  // planes here.  sp_in / sp_out: the seed of every query as a point (x, y, z, target index) in
  // the same order: the match of the previous sweep, written by that sweep.  best_m: the
  // results in scan order (seeds of a following sweep that does not continue the chain).
  constexpr int NQ = 64 / S;
  const int lane = threadIdx.x & 63;
  // the S lanes of a query are ADJACENT lanes: they read S consecutive targets (one or two
  // cache lines per query and load instruction instead of one line per lane)
  const int slice = lane & (S - 1);
  const int wave_id = block * (ICPK_GRID_BLOCK / 64) + (threadIdx.x >> 6);
  const int ip = wave_id * NQ + lane / S;
  GRID_STAMP(0);
  const bool live = ip < nq;
  // (query and seed are asked for BEFORE the loop state is looked at: the state's scalar loads and these two are cold
  // round trips that can run side by side; after the loop's end the two loads were for nothing)
  const float4 q4 = live ? qm4[ip] : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 s4 = live ? sp_in[ip] : make_float4(0.f, 0.f, 0.f, 0.f);
  // (... as is the grid's geometry, wanted only after the seed distance is known; and the state is read in ONE go --
  // flags, translation, rotation -- not flag by flag behind branches: three dependent scalar round trips became one)
  const GridInfo g = *gi;
  asm volatile("" ::"s"(g.lo[0]), "s"(g.lo[1]), "s"(g.lo[2]), "s"(g.inv_h), "s"(g.h), "s"(g.inv_hx), "s"(g.xdiv), "s"(g.nx), "s"(g.ny),
               "s"(g.nz));
  bool apply_rt = false, stop_after = false;
  double Rd[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0};
  float t0 = 0.f, t1 = 0.f, t2 = 0.f;
  if (st) {
    const int done = st->done, stop = st->stop_after_transform, iters = st->iterations;
    t0 = st->rt.t[0], t1 = st->rt.t[1], t2 = st->rt.t[2];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rd[k] = st->Rd[k];
    // (all of them HERE: left alone the compiler sinks each load behind the branch that first needs it)
    asm volatile("" ::"s"(done), "s"(stop), "s"(iters), "s"(t0), "s"(t1), "s"(t2), "s"(Rd[0]), "s"(Rd[1]), "s"(Rd[2]), "s"(Rd[3]),
                 "s"(Rd[4]), "s"(Rd[5]), "s"(Rd[6]), "s"(Rd[7]), "s"(Rd[8]));
    if (done) return;
    stop_after = stop != 0;
    apply_rt = iters > 0 || stop_after;
  }
  const int i = __float_as_int(q4.w);
  float qx = q4.x, qy = q4.y, qz = q4.z;
  if (apply_rt) {  // pointcloud.cpp:321-359: p <- fl32(fl32(R p) + t)
    // (the rotation as the float64 values the step kernel widened once: wave-uniform scalar operands, no per-lane
    // conversions; the same numbers, hence the same bits)
    const double px = qx, py = qy, pz = qz;
    qx = (float)__builtin_fma(Rd[2], pz, __builtin_fma(Rd[1], py, Rd[0] * px)) + t0;
    qy = (float)__builtin_fma(Rd[5], pz, __builtin_fma(Rd[4], py, Rd[3] * px)) + t1;
    qz = (float)__builtin_fma(Rd[8], pz, __builtin_fma(Rd[7], py, Rd[6] * px)) + t2;
    if (live && slice == 0) {
      qm4[ip] = make_float4(qx, qy, qz, q4.w);
      if constexpr (!REC) {
        qxp[i] = qx;
        qyp[i] = qy;
        qzp[i] = qz;
      }
    }
    if (stop_after) return;  // < min_pairs fallback: the motion is applied, no further search
  }
  int bj = __float_as_int(s4.w);
  float bx = s4.x, by = s4.y, bz = s4.z;
  float bd = pair_dist(qx, qy, qz, bx, by, bz);
  if (!(bd <= 3.402823466e38f)) {  // inf/NaN: the reference's literal seed, element 0
    bj = 0;
    bx = oxp[0];
    by = oyp[0];
    bz = ozp[0];
    bd = pair_dist(qx, qy, qz, bx, by, bz);
  }
  float T = filt_threshold(bd);
  GRID_STAMP(1);

  // a NaN best distance can never be replaced (d < NaN and d == NaN are false): no scan
  const bool scan = live && bd == bd;
  bool own = slice == 0;  // this lane's (bx, by, bz) is the point of the group's agreed key (see share)
  // The S lanes of a query share the candidates of a cell box evenly.  Rows are taken S at a
  // time: lane k fetches the range of row k, a prefix sum over the S lanes numbers the
  // candidates of the chunk 0 .. C-1, and lane k evaluates candidates k, k + S, k + 2S, ...,
  // GB of them per round trip, whatever row they are in.  So a chunk costs one trip for its
  // ranges and ceil(C / (S GB)) trips for its candidates, however unevenly the rows are filled.
#ifndef ICPK_GRID_GB
#define ICPK_GRID_GB 5  // (with cells 4x finer along x a query has ~32 candidates: 5 per lane and trip; 64 VGPRs -> 8 waves/SIMD in the steady kernels)
#endif
  constexpr int GB = ICPK_GRID_GB;
  __shared__ int2 rowtab[64 + 8];  // (+ padding, see the byte count below) per query and row of the chunk: (inclusive prefix, start - exclusive prefix)
  int2* const tab = &rowtab[lane & ~(S - 1)];
  auto scan_cells = [&](int x0, int x1, int y0, int y1, int z0, int z1) {
    const int nyr = y1 - y0 + 1;
    const int nrows = scan ? nyr * (z1 - z0 + 1) : 0;

    if (slice == 0) GRID_COUNT(5, nrows);
    for (int r0 = 0; r0 < nrows; r0 += S) {
      const int row = r0 + slice;
      int s0 = 0, len = 0;
      if (row < nrows) {
        // row = rz * nyr + ry without an integer division (~30 instructions): the float quotient is within 1 of the
        // true one (row < 2^20 rows, relative error of rcp and product < 2^-21), two compare-and-fix steps make it exact
        int rz = (int)((float)row * __builtin_amdgcn_rcpf((float)nyr));
        int ry = row - rz * nyr;
        if (ry < 0) {
          rz -= 1;
          ry += nyr;
        } else if (ry >= nyr) {
          rz += 1;
          ry -= nyr;
        }
        const int iy = y0 + ry, iz = z0 + rz;
        const int base = (iz * g.ny + iy) * g.nx;
        int xa = x0, xb = x1;
        // the row's x range trimmed to the BALL of the current best distance (see ball_trim above).  B bounds
        // the real distance |q - t| of every target that could tie or beat bd; its square must be a normal,
        // finite float for the arithmetic to mean anything -- else the cube alone.  (Everything is recomputed
        // per chunk: bd may have improved, and nothing stays live across the candidate loop.)
        const float B = __builtin_fmaf(bd, 1.0f + 0x1p-19f, 0x1p-74f);
        if (GRID_BALL_TRIM && B >= 0x1p-40f && B <= 0x1p60f) {
          const float B2 = B * B * (1.0f + 0x1p-22f);
          float qx_ = qx, qy_ = qy, qz_ = qz;  // (opaque copies: keep the loop-invariant terms from being hoisted into registers)
          asm volatile("" : "+v"(qx_), "+v"(qy_), "+v"(qz_));
          const float slop_y = (__builtin_fabsf(qy_) + (__builtin_fabsf(g.lo[1]) + (float)g.ny * g.h)) * 0x1p-20f;
          const float slop_z = (__builtin_fabsf(qz_) + (__builtin_fabsf(g.lo[2]) + (float)g.nz * g.h)) * 0x1p-20f;
          const float gy = axis_gap(qy_, g.lo[1], g.h, iy, g.ny, slop_y);
          const float gz = axis_gap(qz_, g.lo[2], g.h, iz, g.nz, slop_z);
          const float G2 = __builtin_fmaf(gy, gy, gz * gz) * (1.0f - 0x1p-22f);
          const float dx2 = (B2 - G2) * (1.0f + 0x1p-22f);
          if (dx2 < 0.f) {
            xb = xa - 1;  // no target of this row can tie or beat the current best
          } else {
            // an UPPER bound of the root is all that is wanted: the bare v_sqrt_f32 (within 1 ulp for a normal argument;
            // the correctly rounded sqrtf costs 17 instructions) of an argument kept normal (B >= 2^-40, so 2^-50 is
            // far below anything that matters), times 1 + 2^-19 instead of 1 + 2^-20
            const float rrx = __builtin_fmaf(__builtin_amdgcn_sqrtf(__builtin_fmaxf(dx2, 0x1p-100f)), 1.0f + 0x1p-19f,
                                             __builtin_fmaf(__builtin_fabsf(qx_), 0x1p-21f, 0x1p-74f));
            xa = max(xa, grid_cell(qx_ - rrx, g.lo[0], g.inv_hx, g.nx));
            xb = min(xb, grid_cell(qx_ + rrx, g.lo[0], g.inv_hx, g.nx));
          }
        }
        if (xa <= xb) {
          s0 = cell_start[base + xa];
          len = cell_start[base + xb + 1] - s0;
        }
      }
      GRID_COUNT(6, len);
      // inclusive prefix and total over the S adjacent lanes: DPP moves inside the row of 16
      // (the S lanes of a query are active or inactive together, so every source lane is live)
      int P = len, C = len;
      if constexpr (S >= 2) {
        const int o = dpp_mov<0x111>(P);  // row_shr:1
        if (slice >= 1) P += o;
        C += xor_lane<1>(C);
      }
      if constexpr (S >= 4) {
        const int o = dpp_mov<0x112>(P);  // row_shr:2
        if (slice >= 2) P += o;
        C += xor_lane<2>(C);
      }
      if constexpr (S >= 8) {
        const int o = dpp_mov<0x114>(P);  // row_shr:4
        if (slice >= 4) P += o;
        C += xor_lane<4>(C);
      }
      __builtin_amdgcn_wave_barrier();  // the previous chunk's table has been read
      tab[slice] = make_int2(P, s0 - (P - len));
      __builtin_amdgcn_wave_barrier();  // (one wave per workgroup: LDS operations complete in order)
      // Which row candidate number v lies in = how many of the S inclusive prefixes are <= v.  While every chunk of
      // the wave has at most 127 candidates (the steady state: ~30) the prefixes fit a byte each, all S of them sit
      // in one or two registers (their order does not matter for a count), and ((v | 0x80) - P) has its top bit set
      // exactly in the bytes with P <= v: two subtractions, two ANDs and two population counts per candidate instead
      // of a chain of S compare-and-select pairs.  Same row, same address.
      unsigned W0 = 0, W1 = 0;
      bool small = false;
      if constexpr (S == 4 || S == 8) {
        small = __builtin_amdgcn_ballot_w64(C > 127) == 0;
        if (small) {
          W0 = (unsigned)P;
          W0 |= (unsigned)dpp_mov<0xB1>((int)W0) << 8;   // quad_perm [1,0,3,2]
          W0 |= (unsigned)dpp_mov<0x4E>((int)W0) << 16;  // quad_perm [2,3,0,1]: the quad's four prefixes
          if constexpr (S == 8) W1 = (unsigned)dpp_mov<0x141>((int)W0);  // row_half_mirror: the other quad's
        }
      }
#if defined(ICPK_GRID_STAMPS) && !defined(ICPK_GRID_COUNTS)
      if (C > -1 && r0 == 0 && grid_dbg[wave_id * 8 + 5] == 0) GRID_STAMP(5);  // ranges have arrived
#endif
      for (int v0 = slice; v0 < C; v0 += S * GB) {
        int pk[GB];
        if (small) {
          unsigned vr = (unsigned)v0 * 0x01010101u + 0x80808080u;
#pragma unroll
          for (int k = 0; k < GB; ++k) {
            int r = bcnt_acc((vr - W0) & 0x80808080u, 0);
            if constexpr (S == 8) r = bcnt_acc((vr - W1) & 0x80808080u, r);
            pk[k] = tab[r].y + v0 + k * S;  // (r == S only for v >= C: a lane without a candidate; the entry read is padding)
            vr += (unsigned)S * 0x01010101u;
          }
        } else {
          int2 rt = tab[0];
#pragma unroll
          for (int k = 0; k < GB; ++k) pk[k] = rt.y + v0 + k * S;
#pragma unroll
          for (int t = 1; t < S; ++t) {
            const int pprev = rt.x;
            rt = tab[t];
#pragma unroll
            for (int k = 0; k < GB; ++k) pk[k] = v0 + k * S >= pprev ? rt.y + v0 + k * S : pk[k];
          }
        }
        float4 v[GB];
#pragma unroll
        for (int k = 0; k < GB; ++k) {
          // (lanes without a candidate issue no load; what their registers hold is never looked at -- the test below
          // repeats v < C -- so they are left as they are instead of being cleared: four moves per candidate)
          asm volatile("" : "=v"(v[k].x), "=v"(v[k].y), "=v"(v[k].z), "=v"(v[k].w));
          if (v0 + k * S < C) v[k] = *(const float4*)((const char*)t4 + ((unsigned)pk[k] << 4));  // (< 2^28 targets: icpk_set_target)
        }
#if defined(ICPK_GRID_STAMPS) && !defined(ICPK_GRID_COUNTS)
        if (v[GB - 1].x == v[GB - 1].x && r0 == 0 && grid_dbg[wave_id * 8 + 6] == 0) GRID_STAMP(6);  // first batch has arrived
#endif
#pragma unroll
        for (int k = 0; k < GB; ++k) {
          const float dx = qx - v[k].x, dy = qy - v[k].y, dz = qz - v[k].z;
          const float e2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
          // (the current best itself -- in steady state usually the seed, met again in its cell --
          // cannot improve on itself: skipping it keeps more waves out of the float64 path)
          if (e2 <= T && v0 + k * S < C && __float_as_int(v[k].w) != bj) {
            const float d = pair_dist_seq(dx, dy, dz);
            const int jj = __float_as_int(v[k].w);
            const bool up = (d < bd) | ((d == bd) & (jj < bj));
            bd = up ? d : bd;
            bj = up ? jj : bj;
            bx = up ? v[k].x : bx;
            by = up ? v[k].y : by;
            bz = up ? v[k].z : bz;
            T = up ? filt_threshold(d) : T;
            own = own | up;
          }
        }
      }
    }
  };
  // Expanding search: pass k looks at the cells within R = 1, 2, 4, ... of the query's own
  // cell (clipped to the cube of the current best distance), the S lanes share what they
  // found, and the search ends as soon as the cube of the shared best distance lies inside the
  // box just scanned -- then every target that could tie or beat it has been examined.  With a
  // good seed that is the first pass; a loose or far seed costs work proportional to the true
  // NN distance, not to the seed's.  EXPAND = false (seeds are the previous sweep's matches):
  // one pass over the whole cube of the seed distance -- fewer round trips than two passes,
  // and a leaner kernel.
  // The S lanes of a query agree on the best (distance, index) so far: two min-reductions over the group --
  // distance bits (non-negative floats order like their bit patterns), then the index among the lanes that hold
  // that distance.  The matched POINT does not travel: `own` marks the one lane whose (bx, by, bz) belong to the
  // agreed key -- the lane that found it, or slice 0 for the seed all S lanes start from -- and that lane stores the
  // result.  (There always is one: a lane holds a key either as its own find or as a copy of a lane that still holds
  // it, unless that lane has found something better since, and then the better key is the agreed one.  Finds are
  // unique to a lane: the S lanes walk disjoint candidates and skip the current best itself.)
  auto group_min = [&](unsigned v) -> unsigned {  // all-reduce over the S adjacent lanes (any pairing pattern does for a minimum)
    if constexpr (S >= 2) v = min(v, (unsigned)dpp_mov<0xB1>((int)v));   // quad_perm [1,0,3,2]
    if constexpr (S >= 4) v = min(v, (unsigned)dpp_mov<0x4E>((int)v));   // quad_perm [2,3,0,1]
    if constexpr (S >= 8) v = min(v, (unsigned)dpp_mov<0x141>((int)v));  // row_half_mirror: lane i <-> 7 - i, i.e. the other quad
    return v;
  };
  auto share = [&]() {
    if (S > 1) {
      const unsigned db = __float_as_uint(bd);
      const unsigned dm = group_min(db);
      const unsigned jm = group_min(db == dm ? (unsigned)bj : 0xffffffffu);
      if (scan) {  // (a NaN distance is not ordered by its bits; such lanes never scan, and the S lanes of a query scan or do not together)
        own = own && db == dm && (unsigned)bj == jm;
        bd = __uint_as_float(dm);
        bj = (int)jm;
        T = filt_threshold(bd);
      }
    }
  };
  GRID_STAMP(2);
  if constexpr (!EXPAND) {
    int x0, x1, y0, y1, z0, z1;
    cube_cells(qx, bd, g.lo[0], g.inv_hx, g.nx, x0, x1);
    cube_cells(qy, bd, g.lo[1], g.inv_h, g.ny, y0, y1);
    cube_cells(qz, bd, g.lo[2], g.inv_h, g.nz, z0, z1);
    scan_cells(x0, x1, y0, y1, z0, z1);
    share();
  } else {
    const int cx = grid_cell(qx, g.lo[0], g.inv_hx, g.nx);
    const int cy = grid_cell(qy, g.lo[1], g.inv_h, g.ny);
    const int cz = grid_cell(qz, g.lo[2], g.inv_h, g.nz);
    bool active = scan;
    for (int R = 1; __builtin_amdgcn_ballot_w64(active) != 0; R = R < (1 << 20) ? 2 * R : R) {
      int X0, X1, Y0, Y1, Z0, Z1;
      cube_cells(qx, bd, g.lo[0], g.inv_hx, g.nx, X0, X1);
      cube_cells(qy, bd, g.lo[1], g.inv_h, g.ny, Y0, Y1);
      cube_cells(qz, bd, g.lo[2], g.inv_h, g.nz, Z0, Z1);
      const int x0 = max(X0, cx - R * g.xdiv), x1 = min(X1, cx + R * g.xdiv);  // the same physical reach on every axis
      const int y0 = max(Y0, cy - R), y1 = min(Y1, cy + R);
      const int z0 = max(Z0, cz - R), z1 = min(Z1, cz + R);
      if (active) scan_cells(x0, x1, y0, y1, z0, z1);
      share();
      cube_cells(qx, bd, g.lo[0], g.inv_hx, g.nx, X0, X1);
      cube_cells(qy, bd, g.lo[1], g.inv_h, g.ny, Y0, Y1);
      cube_cells(qz, bd, g.lo[2], g.inv_h, g.nz, Z0, Z1);
      active = active && !(X0 >= x0 && X1 <= x1 && Y0 >= y0 && Y1 <= y1 && Z0 >= z0 && Z1 <= z1);
    }
  }

  GRID_STAMP(3);
  // (the S lanes of a query already agree on the key: the last pass ended with a merge; the lane that owns the
  // matched point stores)
  const nn_key_t key = ((nn_key_t)__float_as_uint(bd) << 32) | (nn_key_t)(unsigned)bj;
  if (live && own) {
    const float4 m4 = make_float4(bx, by, bz, __int_as_float((int)(unsigned)(key & 0xffffffffu)));
    if constexpr (REC) {
      rec[2 * (size_t)i] = make_float4(qx, qy, qz, __uint_as_float((unsigned)(key >> 32)));
      rec[2 * (size_t)i + 1] = m4;
    } else {
      best[i] = key;
      best_m[ip] = key;
    }
    sp_out[ip] = m4;
  }
  GRID_STAMP(4);
}

template <int S, bool EXPAND, bool REC>
__global__ __launch_bounds__(ICPK_GRID_BLOCK) void nn_grid_kernel(const GridSweepArgs a) {
  nn_grid_body<S, EXPAND, REC>(a.qx, a.qy, a.qz, a.nq, a.qm4, a.t4, a.cell_start, a.gi, a.ox, a.oy, a.oz, a.sp_in,
                               a.sp_out, a.best, a.best_m, a.rec, a.st, blockIdx.x);
}

// frame-batch mode: blockIdx.y = pair.  The pairs of a group differ in size: workgroups beyond
// a pair's own count leave at once.
template <int S, bool EXPAND>
__global__ __launch_bounds__(ICPK_GRID_BLOCK) void nn_grid_batch_kernel(const GridSweepBatch b) {
  const GridSweepArgs& a = b.p[blockIdx.y];
  if ((long long)blockIdx.x * (64 / S) >= a.nq) return;
  nn_grid_body<S, EXPAND, true>(a.qx, a.qy, a.qz, a.nq, a.qm4, a.t4, a.cell_start, a.gi, a.ox, a.oy, a.oz, a.sp_in,
                                a.sp_out, a.best, a.best_m, a.rec, a.st, blockIdx.x);
}

static inline int grid_blocks(int nq, int slices) { return (nq + (64 / slices) - 1) / (64 / slices); }

void launch_nn_grid(const GridSweepArgs& a, int slices, int expand, hipStream_t s) {
#define ICPK_LAUNCH3(SL, EX, RC) \
  hipLaunchKernelGGL((nn_grid_kernel<SL, EX, RC>), dim3(grid_blocks(a.nq, SL)), dim3(ICPK_GRID_BLOCK), 0, s, a)
#define ICPK_LAUNCH2(SL, EX)      \
  do {                            \
    if (a.rec) {                  \
      ICPK_LAUNCH3(SL, EX, true);  \
    } else {                      \
      ICPK_LAUNCH3(SL, EX, false); \
    }                             \
  } while (0)
#define ICPK_LAUNCH(SL)    \
  do {                     \
    if (expand) {          \
      ICPK_LAUNCH2(SL, true);  \
    } else {               \
      ICPK_LAUNCH2(SL, false); \
    }                      \
  } while (0)
  switch (slices) {
    case 1: ICPK_LAUNCH(1); break;
    case 2: ICPK_LAUNCH(2); break;
    case 8: ICPK_LAUNCH(8); break;
    default: ICPK_LAUNCH(4); break;
  }
#undef ICPK_LAUNCH
#undef ICPK_LAUNCH2
#undef ICPK_LAUNCH3
}

void launch_nn_grid_batch(const GridSweepBatch& b, int count, int slices, int expand, hipStream_t s) {
  int nq_max = 0;
  for (int k = 0; k < count; ++k) nq_max = b.p[k].nq > nq_max ? b.p[k].nq : nq_max;
  if (count <= 0 || nq_max <= 0) return;
#define ICPK_LAUNCH2(SL, EX)                                                                            \
  hipLaunchKernelGGL((nn_grid_batch_kernel<SL, EX>), dim3(grid_blocks(nq_max, SL), count), dim3(ICPK_GRID_BLOCK), 0, s, b)
#define ICPK_LAUNCH(SL)    \
  do {                     \
    if (expand) {          \
      ICPK_LAUNCH2(SL, true);  \
    } else {               \
      ICPK_LAUNCH2(SL, false); \
    }                      \
  } while (0)
  switch (slices) {
    case 1: ICPK_LAUNCH(1); break;
    case 2: ICPK_LAUNCH(2); break;
    case 4: ICPK_LAUNCH(4); break;
    default: ICPK_LAUNCH(8); break;
  }
#undef ICPK_LAUNCH
#undef ICPK_LAUNCH2
}

// queries / seeds of a sweep that does not continue a chain of grid sweeps: Morton-ordered
// copies from the caller's planes and from the seed keys
__global__ void grid_query_points_kernel(const float* __restrict__ qx, const float* __restrict__ qy,
                                         const float* __restrict__ qz, const int* __restrict__ qperm, int nq,
                                         const nn_key_t* __restrict__ seed_m, const float* __restrict__ ox,
                                         const float* __restrict__ oy, const float* __restrict__ oz,
                                         float4* __restrict__ qm4, float4* __restrict__ sp) {
  const int ip = blockIdx.x * blockDim.x + threadIdx.x;
  if (ip >= nq) return;
  const int i = qperm[ip];
  qm4[ip] = make_float4(qx[i], qy[i], qz[i], __int_as_float(i));
  const int js = (int)(unsigned)(seed_m[ip] & 0xffffffffu);
  sp[ip] = make_float4(ox[js], oy[js], oz[js], __int_as_float(js));
}

void launch_grid_query_points(const float* qx, const float* qy, const float* qz, const int* qperm, int nq,
                              const nn_key_t* seed_m, const float* ox, const float* oy, const float* oz, float4* qm4,
                              float4* sp, hipStream_t s) {
  if (nq <= 0) return;
  hipLaunchKernelGGL(grid_query_points_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, qx, qy, qz, qperm, nq, seed_m,
                     ox, oy, oz, qm4, sp);
}

// After a loop of REC sweeps: the caller-order views the sweeps did not keep current -- the moved source planes from
// the scan-order queries (thread t as scan position) and the (distance, index) keys from the records (thread t as
// query index).
__global__ void grid_unpack_kernel(const float4* __restrict__ qm4, const float4* __restrict__ rec, int nq,
                                   float* __restrict__ qx, float* __restrict__ qy, float* __restrict__ qz,
                                   nn_key_t* __restrict__ best) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nq) return;
  const float4 q = qm4[t];
  const int i = __float_as_int(q.w);
  qx[i] = q.x;
  qy[i] = q.y;
  qz[i] = q.z;
  const float d = rec[2 * (size_t)t].w;
  const float j = rec[2 * (size_t)t + 1].w;
  best[t] = ((nn_key_t)__float_as_uint(d) << 32) | (nn_key_t)__float_as_uint(j);
}
void launch_grid_unpack(const float4* qm4, const float4* rec, int nq, float* qx, float* qy, float* qz, nn_key_t* best,
                        hipStream_t s) {
  if (nq <= 0) return;
  hipLaunchKernelGGL(grid_unpack_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, qm4, rec, nq, qx, qy, qz, best);
}

}  // namespace icpk
