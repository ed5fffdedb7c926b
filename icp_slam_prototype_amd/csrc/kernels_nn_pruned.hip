// kernels_nn_pruned.hip -- K1c: exact NN with bounding-box pruning (ICPK_NN_PRUNED).
//
// Same result as nn_exact_kernel / nn_filtered_kernel (tests check all three against
// the oracle bit for bit).  The target cloud is scanned in Morton order (kernels_sort.hip)
// so that every 1024-point tile and 128-point sub-tile is a compact cluster with a tight
// bounding box; the queries are taken in Morton order too, so the 64 queries of a wave
// form a small cluster themselves.
//
// One wave64 per workgroup -- no barriers, no atomics, one plain store per query:
//   1. init: query, seed candidate (previous match), exact seed distance d_b, filter
//      threshold T(d_b) per lane (as in nn_filtered_kernel);
//   2. wave bounding box of the queries (xor-butterfly min/max) and T_max;
//   3. tile pass: LANES RANGE OVER TILES -- each lane tests one tile's box against the
//      wave box (coalesced box loads, one ballot for 64 tiles), passing tiles are
//      compacted into LDS with mbcnt; likewise 8 tiles x 8 sub-tiles per round;
//   4. each passing sub-tile: its 128 targets are staged once through LDS (coalesced
//      loads, next sub-tile prefetched in registers), broadcast ds_reads feed the fp32
//      filter (group_estimates) and the rare exact re-evaluation (group_exact).
// A box is skipped only if it cannot contain a target that passes any lane's filter
// (see box test below), so pruning never changes the result.
#include "icpk_internal.h"
#include "nn_device.h"

namespace icpk {

constexpr int NP_MAX_SUBS = 16 * NN_SUBS;  // candidate sub-tiles of one chunk of 16 passing tiles

// Diagnostic build only (tools/stamp_pruned.py compiles with -DICPK_NP_STAMPS): per-wave
// s_memtime stamps of the phases, written to a buffer nothing else reads.
#ifdef ICPK_NP_STAMPS
__device__ unsigned long long np_dbg[8 * 4096];
#define NP_STAMP(k)                                                                  \
  do {                                                                               \
    if (threadIdx.x == 0 && blockIdx.x < 4096) np_dbg[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define NP_COUNT(k, v)                                                               \
  do {                                                                               \
    if (threadIdx.x == 0 && blockIdx.x < 4096) np_dbg[blockIdx.x * 8 + (k)] += (v);  \
  } while (0)
extern "C" int icpk_debug_read_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(np_dbg), sizeof(np_dbg));
}
extern "C" int icpk_debug_clear_stamps() {
  static unsigned long long zero[8 * 4096];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(np_dbg), zero, sizeof(zero));
}
#else
#define NP_STAMP(k)
#define NP_COUNT(k, v)
#endif

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = __builtin_fminf(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}

// squared gap between box [lo,hi] and the wave's query box [alo,ahi]; a lower bound of
// every query-target squared distance (fp32 rounding covered by the 2^-19 margin at
// the comparison).  Empty boxes (lo = +inf, hi = -inf) give +inf.
__device__ __forceinline__ float box_gap2(float lox, float loy, float loz, float hix, float hiy, float hiz,
                                          const float (&alo)[3], const float (&ahi)[3]) {
  const float dx = __builtin_fmaxf(__builtin_fmaxf(lox - ahi[0], alo[0] - hix), 0.f);
  const float dy = __builtin_fmaxf(__builtin_fmaxf(loy - ahi[1], alo[1] - hiy), 0.f);
  const float dz = __builtin_fmaxf(__builtin_fmaxf(loz - ahi[2], alo[2] - hiz), 0.f);
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// S = target slices per query: a wave holds 64/S queries, each replicated on S lanes that
// scan different quarters (S = 4) of every candidate sub-tile.  More, shorter waves: with
// ~92k queries S = 1 gives only ~1.4 waves per SIMD (issue- and latency-bound), S = 4 gives
// ~5.6.  The S partial results of a query are merged lexicographically by xor-shuffles.
template <int S>
__global__ __launch_bounds__(64) void nn_pruned_kernel(
    float* __restrict__ qxp, float* __restrict__ qyp, float* __restrict__ qzp, int nq,
    const float* __restrict__ txp, const float* __restrict__ typ, const float* __restrict__ tzp, int ntiles,
    const float* __restrict__ oxp, const float* __restrict__ oyp, const float* __restrict__ ozp,
    const int* __restrict__ tperm, const int* __restrict__ qperm, const float* __restrict__ tbox, int tbox_stride,
    const float* __restrict__ sbox, int sbox_stride, const nn_key_t* __restrict__ seed_m,
    nn_key_t* __restrict__ best, nn_key_t* __restrict__ best_m, int recheck, const LoopState* __restrict__ st) {
  // seed_m / best_m: seeds and results in QUERY MORTON ORDER (position ip), so the seed
  // look-up does not wait for the qperm gather; best: results in the caller's order.
  // recheck: re-test every candidate against the current thresholds just before it is
  // scanned (pays off in the first sweep of an alignment, when the seeds are loose).
  // st (device-side loop only): besides the stop flags it carries the rigid transform of
  // the iteration that just ended; K3 is fused here -- every query belongs to exactly one
  // wave, which moves it (same arithmetic as transform_kernel), writes it back for K2 and
  // searches from the new position.
  bool apply_rt = false, stop_after = false;
  if (st) {
    if (st->done) return;
    stop_after = st->stop_after_transform != 0;
    apply_rt = st->iterations > 0 || stop_after;
  }
  constexpr int NQ = 64 / S;              // queries per wave
  constexpr int SLICE = NN_SUB / S;       // targets of a sub-tile scanned by one lane
  static_assert(SLICE % NNF_G == 0, "slice must be whole groups");
  __shared__ int tile_list[64];
  __shared__ int sub_list[NP_MAX_SUBS];
  __shared__ float sub_box[6][NP_MAX_SUBS];  // boxes of the candidates, for the per-lane re-test
  __shared__ __attribute__((aligned(16))) float stage[3][NN_SUB];

  NP_STAMP(0);
  const int lane = threadIdx.x;
  const int slice = lane / NQ;
  const int ip = blockIdx.x * NQ + (lane % NQ);
  const bool live = ip < nq;
  const int i = live ? qperm[ip] : 0;
  float qx[1], qy[1], qz[1], bd[1], T[1];
  int bj[1];
  qx[0] = live ? qxp[i] : 0.f;
  qy[0] = live ? qyp[i] : 0.f;
  qz[0] = live ? qzp[i] : 0.f;
  if (apply_rt) {  // pointcloud.cpp:321-359: p <- fl32(fl32(R p) + t)
    const Rt rt = st->rt;
    const double px = qx[0], py = qy[0], pz = qz[0];
    qx[0] = (float)__builtin_fma((double)rt.R[2], pz, __builtin_fma((double)rt.R[1], py, (double)rt.R[0] * px)) + rt.t[0];
    qy[0] = (float)__builtin_fma((double)rt.R[5], pz, __builtin_fma((double)rt.R[4], py, (double)rt.R[3] * px)) + rt.t[1];
    qz[0] = (float)__builtin_fma((double)rt.R[8], pz, __builtin_fma((double)rt.R[7], py, (double)rt.R[6] * px)) + rt.t[2];
    if (live && slice == 0) {
      qxp[i] = qx[0];
      qyp[i] = qy[0];
      qzp[i] = qz[0];
    }
    if (stop_after) return;  // < min_pairs fallback: the motion is applied, no further search
  }
  {
    int js = live ? (int)(unsigned)(seed_m[ip] & 0xffffffffu) : 0;
    float ds = pair_dist(qx[0], qy[0], qz[0], oxp[js], oyp[js], ozp[js]);
    if (!(ds <= 3.402823466e38f)) {  // inf/NaN: the reference's literal seed, element 0
      js = 0;
      ds = pair_dist(qx[0], qy[0], qz[0], oxp[0], oyp[0], ozp[0]);
    }
    bd[0] = ds;
    bj[0] = js;
    T[0] = live ? filt_threshold(ds) : -1.f;  // dead lanes never pass
  }

  // wave box of the live queries
  float alo[3], ahi[3];
  alo[0] = wave_min(live ? qx[0] : __builtin_inff());
  alo[1] = wave_min(live ? qy[0] : __builtin_inff());
  alo[2] = wave_min(live ? qz[0] : __builtin_inff());
  ahi[0] = wave_max(live ? qx[0] : -__builtin_inff());
  ahi[1] = wave_max(live ? qy[0] : -__builtin_inff());
  ahi[2] = wave_max(live ? qz[0] : -__builtin_inff());
  // a NaN query leaves the box untouched but must see every target
  const bool weird = live && !(qx[0] - qx[0] == 0.f && qy[0] - qy[0] == 0.f && qz[0] - qz[0] == 0.f);
  const bool any_weird = __builtin_amdgcn_ballot_w64(weird) != 0;
  NP_STAMP(1);  // init + wave box done

  auto survives = [&](int s) -> bool {
    const bool h = any_weird || box_may_hit<1>(qx, qy, qz, sub_box[0][s], sub_box[1][s], sub_box[2][s],
                                               sub_box[3][s], sub_box[4][s], sub_box[5][s], T);
    return __builtin_amdgcn_ballot_w64(h) != 0;
  };

  for (int tb = 0; tb < ntiles; tb += 64) {
    float tmax = any_weird ? __builtin_inff() : wave_max(T[0]) * (1.0f + 0x1p-19f);
    // ---- tile pass: lane <-> tile ----
    const int t = tb + lane;
    bool tp = false;
    if (t < ntiles)
      tp = box_gap2(tbox[t], tbox[tbox_stride + t], tbox[2 * tbox_stride + t], tbox[3 * tbox_stride + t],
                    tbox[4 * tbox_stride + t], tbox[5 * tbox_stride + t], alo, ahi) <= tmax;
    const unsigned long long tm = __builtin_amdgcn_ballot_w64(tp);
    if (tm == 0) continue;
    const int ntl = __popcll(tm);
    __syncthreads();  // the previous round's lists are no longer read
    if (tp) tile_list[__popcll(tm & ((1ull << lane) - 1ull))] = t;
    __syncthreads();
    // ---- passing tiles, 16 at a time (<= 128 candidate sub-tiles in LDS) ----
    for (int g0 = 0; g0 < ntl; g0 += 16) {
      if (g0) tmax = any_weird ? __builtin_inff() : wave_max(T[0]) * (1.0f + 0x1p-19f);
      int nsl = 0;
      __syncthreads();  // candidates of the previous chunk fully consumed
#pragma unroll
      for (int h = 0; h < 2; ++h) {  // sub-tile pass: lane <-> (tile, sub-tile), 8 x 8 per step
        const int ti = g0 + 8 * h + (lane >> 3);
        bool sp = false;
        int sb = 0;
        if (ti < ntl && ti < g0 + 16) {
          sb = tile_list[ti] * NN_SUBS + (lane & 7);
          sp = box_gap2(sbox[sb], sbox[sbox_stride + sb], sbox[2 * sbox_stride + sb], sbox[3 * sbox_stride + sb],
                        sbox[4 * sbox_stride + sb], sbox[5 * sbox_stride + sb], alo, ahi) <= tmax;
        }
        const unsigned long long sm = __builtin_amdgcn_ballot_w64(sp);
        if (sp) {
          const int w = nsl + __popcll(sm & ((1ull << lane) - 1ull));
          sub_list[w] = sb;
#pragma unroll
          for (int c = 0; c < 6; ++c) sub_box[c][w] = sbox[c * sbox_stride + sb];
        }
        nsl += __popcll(sm);
      }
      __syncthreads();
      NP_COUNT(5, nsl);  // coarse candidates
      if (nsl == 0) continue;
      NP_STAMP(2);
      // ---- eager per-lane re-test (lane <-> query again) with the thresholds as they are
      // now: the wave box can be much larger than the union of the queries' own search balls
      // (a Morton run may straddle a jump of the curve).  Candidate boxes come back from LDS
      // by broadcast, 8 per step so the read latencies overlap; survivors compacted in place.
      {
        int keep = 0;
        for (int s0 = 0; s0 < nsl; s0 += 8) {
          unsigned hitbits = 0;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int sc = s0 + k < nsl ? s0 + k : nsl - 1;  // clamp: duplicates are masked below
            const bool h = any_weird || box_may_hit<1>(qx, qy, qz, sub_box[0][sc], sub_box[1][sc], sub_box[2][sc],
                                                       sub_box[3][sc], sub_box[4][sc], sub_box[5][sc], T);
            if (__builtin_amdgcn_ballot_w64(h) != 0 && s0 + k < nsl) hitbits |= 1u << k;
          }
          while (hitbits) {
            const int k = __builtin_ctz(hitbits);
            hitbits &= hitbits - 1;
            const int sc = s0 + k;  // keep <= sc: in place; LDS operations of one wave execute in order
            if (lane < 6) sub_box[lane][keep] = sub_box[lane][sc];
            if (lane == 6) sub_list[keep] = sub_list[sc];
            ++keep;
          }
        }
        nsl = keep;
        __syncthreads();
        NP_COUNT(6, nsl);  // survivors
        NP_STAMP(3);
        if (nsl == 0) continue;
      }
      // ---- scan.  Each candidate is re-tested (lazily) against the CURRENT thresholds just
      // before it is scanned; the next surviving candidate's 128 targets are loaded into
      // registers while the current one is scanned. ----
      auto next_surviving = [&](int s) -> int {
        while (s < nsl && !survives(s)) ++s;
        return s;
      };
      // two register sets (A, B) in ping-pong: while candidate c is scanned the loads of c+1
      // AND c+2 are in flight, so each has two scans of time to land
      struct Regs {
        float x0, x1, y0, y1, z0, z1;
      };
      auto load = [&](Regs& r, int s) {
        const int o = sub_list[s] * NN_SUB + lane;
        r.x0 = txp[o];
        r.x1 = txp[o + 64];
        r.y0 = typ[o];
        r.y1 = typ[o + 64];
        r.z0 = tzp[o];
        r.z1 = tzp[o + 64];
      };
      auto scan = [&](const Regs& r, int c) {
        __syncthreads();  // previous sub-tile fully consumed
        stage[0][lane] = r.x0;
        stage[0][64 + lane] = r.x1;
        stage[1][lane] = r.y0;
        stage[1][64 + lane] = r.y1;
        stage[2][lane] = r.z0;
        stage[2][64 + lane] = r.z1;
        __syncthreads();
      };
      auto consume = [&](int c) {
        const int jbase = sub_list[c] * NN_SUB + slice * SLICE;
        NP_COUNT(7, 1);  // sub-tiles scanned
        const float Told = T[0];
#pragma unroll 2
        for (int g = 0; g < SLICE; g += NNF_G) {
          const int o = slice * SLICE + g;
          float X[NNF_G], Y[NNF_G], Z[NNF_G];
          const float4 xa = *reinterpret_cast<const float4*>(&stage[0][o]);  // one address per slice:
          const float4 xb = *reinterpret_cast<const float4*>(&stage[0][o + 4]);  // LDS broadcast
          const float4 ya = *reinterpret_cast<const float4*>(&stage[1][o]);
          const float4 yb = *reinterpret_cast<const float4*>(&stage[1][o + 4]);
          const float4 za = *reinterpret_cast<const float4*>(&stage[2][o]);
          const float4 zb = *reinterpret_cast<const float4*>(&stage[2][o + 4]);
          X[0] = xa.x; X[1] = xa.y; X[2] = xa.z; X[3] = xa.w; X[4] = xb.x; X[5] = xb.y; X[6] = xb.z; X[7] = xb.w;
          Y[0] = ya.x; Y[1] = ya.y; Y[2] = ya.z; Y[3] = ya.w; Y[4] = yb.x; Y[5] = yb.y; Y[6] = yb.z; Y[7] = yb.w;
          Z[0] = za.x; Z[1] = za.y; Z[2] = za.z; Z[3] = za.w; Z[4] = zb.x; Z[5] = zb.y; Z[6] = zb.z; Z[7] = zb.w;
          float e[NNF_G][1], m[1];
          group_estimates<1>(qx, qy, qz, X, Y, Z, e, m);
          if (__builtin_amdgcn_ballot_w64(m[0] <= T[0]) != 0)
            group_exact_lanes<1>(qx, qy, qz, X, Y, Z, e, jbase + g, tperm, bd, bj, T);
        }
        // the S lanes of a query share the tightest threshold (only when one changed)
        if (S > 1 && __builtin_amdgcn_ballot_w64(T[0] != Told) != 0) {
#pragma unroll
          for (int m = NQ; m < 64; m <<= 1) T[0] = __builtin_fminf(T[0], __shfl_xor(T[0], m, 64));
        }
      };
      // candidate order: as listed, or (first sweep) each judged against the thresholds current
      // at the time its loads are issued
      auto next_cand = [&](int s) -> int { return recheck ? next_surviving(s) : s; };
      Regs A, B;
      int c0 = next_cand(0);
      if (c0 < nsl) load(A, c0);
      int c1 = c0 < nsl ? next_cand(c0 + 1) : nsl;
      if (c1 < nsl) load(B, c1);
      while (c0 < nsl) {
        scan(A, c0);  // stage A
        const int c2 = c1 < nsl ? next_cand(c1 + 1) : nsl;
        if (c2 < nsl) load(A, c2);
        consume(c0);
        if (c1 >= nsl) break;
        scan(B, c1);  // stage B
        const int c3 = c2 < nsl ? next_cand(c2 + 1) : nsl;
        if (c3 < nsl) load(B, c3);
        consume(c1);
        c0 = c2;
        c1 = c3;
      }
    }
  }

  NP_STAMP(4);
  nn_key_t key = ((nn_key_t)__float_as_uint(bd[0]) << 32) | (nn_key_t)(unsigned)bj[0];
  if (S > 1) {  // lexicographic (distance, index) min over the S lanes of the query
#pragma unroll
    for (int m = NQ; m < 64; m <<= 1) {
      const nn_key_t o = __shfl_xor(key, m, 64);
      key = o < key ? o : key;
    }
  }
  if (live && slice == 0) {
    best[i] = key;
    best_m[ip] = key;
  }
}

void launch_nn_pruned(const NnArgs& a, const nn_key_t* seed_m, nn_key_t* best_m, const NnBoxes& b, int slices,
                      int recheck, const LoopState* st, hipStream_t s) {
  const int ntiles = a.nt_pad / NN_TILE;
#define ICPK_LAUNCH(SL)                                                                                            \
  hipLaunchKernelGGL(nn_pruned_kernel<SL>, dim3((a.nq + 64 / SL - 1) / (64 / SL)), dim3(64), 0, s,               \
                     const_cast<float*>(a.qx), const_cast<float*>(a.qy), const_cast<float*>(a.qz),                \
                     a.nq, a.tx, a.ty, a.tz, ntiles, b.ox, b.oy, b.oz, b.tperm, b.qperm, b.tbox, b.tbox_stride,    \
                     b.sbox, b.sbox_stride, seed_m, a.best, best_m, recheck, st)
  switch (slices) {
    case 1: ICPK_LAUNCH(1); break;
    case 2: ICPK_LAUNCH(2); break;
    case 8: ICPK_LAUNCH(8); break;
    case 16: ICPK_LAUNCH(16); break;
    default: ICPK_LAUNCH(4); break;
  }
#undef ICPK_LAUNCH
}

// First-sweep seeds without a brute-force pre-pass: binary search of the query's Morton
// code in the sorted target keys, then the nearest (fp32 distance) of the 2W+1 targets
// around that position in the Morton-ordered planes.  Any index is a valid seed -- it
// only sets the initial search radius -- so nothing here needs to be exact.
constexpr int SEED_W = 8;
__global__ void seed_morton_kernel(const unsigned* __restrict__ qkeys, const int* __restrict__ qperm, int nq,
                                   const float* __restrict__ qxp, const float* __restrict__ qyp,
                                   const float* __restrict__ qzp, const unsigned* __restrict__ tkeys,
                                   const float* __restrict__ sx, const float* __restrict__ sy,
                                   const float* __restrict__ sz, const int* __restrict__ tperm, int nt,
                                   nn_key_t* __restrict__ seed_m) {
  const int ip = blockIdx.x * blockDim.x + threadIdx.x;
  if (ip >= nq) return;
  const int i = qperm[ip];
  const unsigned k = qkeys[i];
  int lo = 0, hi = nt;  // first position with tkeys[pos] >= k
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (tkeys[mid] < k) lo = mid + 1; else hi = mid;
  }
  const float qx = qxp[i], qy = qyp[i], qz = qzp[i];
  int a = lo - SEED_W, b = lo + SEED_W;
  if (a < 0) a = 0;
  if (b > nt - 1) b = nt - 1;
  float bestv = __builtin_inff();
  int bestp = a;
  for (int p = a; p <= b; ++p) {
    const float dx = qx - sx[p], dy = qy - sy[p], dz = qz - sz[p];
    const float e = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    if (e < bestv) {
      bestv = e;
      bestp = p;
    }
  }
  seed_m[ip] = (nn_key_t)(unsigned)tperm[bestp];
}

// seeds from a sweep of another kernel (caller's order) -> query Morton order
__global__ void seed_gather_kernel(const nn_key_t* __restrict__ best, const int* __restrict__ qperm, int nq,
                                   nn_key_t* __restrict__ seed_m) {
  const int ip = blockIdx.x * blockDim.x + threadIdx.x;
  if (ip < nq) seed_m[ip] = best[qperm[ip]];
}

void launch_seed_gather(const nn_key_t* best, const int* qperm, int nq, nn_key_t* seed_m, hipStream_t s) {
  if (nq <= 0) return;
  hipLaunchKernelGGL(seed_gather_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, best, qperm, nq, seed_m);
}

void launch_seed_morton(const unsigned* qkeys, const int* qperm, int nq, const float* qx, const float* qy,
                        const float* qz, const unsigned* tkeys, const float* sx, const float* sy, const float* sz,
                        const int* tperm, int nt, nn_key_t* seed_m, hipStream_t s) {
  if (nq <= 0) return;
  hipLaunchKernelGGL(seed_morton_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, qkeys, qperm, nq, qx, qy, qz, tkeys,
                     sx, sy, sz, tperm, nt, seed_m);
}

}  // namespace icpk
