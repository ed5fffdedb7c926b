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

constexpr int NP_MAX_SUBS = 64 * NN_SUBS;  // passing sub-tiles of one 64-tile round

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

__global__ __launch_bounds__(64) void nn_pruned_kernel(
    const float* __restrict__ qxp, const float* __restrict__ qyp, const float* __restrict__ qzp, int nq,
    const float* __restrict__ txp, const float* __restrict__ typ, const float* __restrict__ tzp, int ntiles,
    const float* __restrict__ oxp, const float* __restrict__ oyp, const float* __restrict__ ozp,
    const int* __restrict__ tperm, const int* __restrict__ qperm, const float* __restrict__ tbox, int tbox_stride,
    const float* __restrict__ sbox, int sbox_stride, const nn_key_t* __restrict__ seed, int seed_scale,
    nn_key_t* __restrict__ best, const int* __restrict__ stop) {
  if (loop_stopped(stop)) return;
  __shared__ int tile_list[64];
  __shared__ int sub_list[NP_MAX_SUBS];
  __shared__ float sub_box[6][NP_MAX_SUBS];  // boxes of the candidates, for the per-lane re-test
  __shared__ __attribute__((aligned(16))) float stage[3][NN_SUB];

  const int lane = threadIdx.x;
  const int ip = blockIdx.x * 64 + lane;
  const bool live = ip < nq;
  const int i = live ? qperm[ip] : 0;
  float qx[1], qy[1], qz[1], bd[1], T[1];
  int bj[1];
  qx[0] = live ? qxp[i] : 0.f;
  qy[0] = live ? qyp[i] : 0.f;
  qz[0] = live ? qzp[i] : 0.f;
  {
    int js = live ? (int)(unsigned)(seed[i] & 0xffffffffu) * seed_scale : 0;
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

  for (int tb = 0; tb < ntiles; tb += 64) {
    const float tmax = any_weird ? __builtin_inff() : wave_max(T[0]) * (1.0f + 0x1p-19f);
    // ---- tile pass: lane <-> tile ----
    const int t = tb + lane;
    bool tp = false;
    if (t < ntiles)
      tp = box_gap2(tbox[t], tbox[tbox_stride + t], tbox[2 * tbox_stride + t], tbox[3 * tbox_stride + t],
                    tbox[4 * tbox_stride + t], tbox[5 * tbox_stride + t], alo, ahi) <= tmax;
    const unsigned long long tm = __builtin_amdgcn_ballot_w64(tp);
    if (tm == 0) continue;
    const int ntl = __popcll(tm);
    if (tp) tile_list[__popcll(tm & ((1ull << lane) - 1ull))] = t;
    __syncthreads();
    // ---- sub-tile pass: 8 tiles x 8 sub-tiles per round ----
    int nsl = 0;
    for (int g = 0; g < ntl; g += 8) {
      const int ti = g + (lane >> 3);
      bool sp = false;
      int sb = 0;
      if (ti < ntl) {
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
    if (nsl == 0) continue;
    // ---- scan.  Before a candidate is scanned it is re-tested per lane (lane <-> query
    // again) against the CURRENT thresholds: the wave box can be much larger than the
    // union of the queries' own search balls (a Morton run may straddle a jump of the
    // curve) and the thresholds tighten as soon as the first few sub-tiles are scanned.
    // Candidate boxes come back from LDS by broadcast.  The next surviving candidate's
    // 128 targets are loaded into registers while the current one is scanned. ----
    auto survives = [&](int s) -> bool {
      const bool h = any_weird || box_may_hit<1>(qx, qy, qz, sub_box[0][s], sub_box[1][s], sub_box[2][s],
                                                 sub_box[3][s], sub_box[4][s], sub_box[5][s], T);
      return __builtin_amdgcn_ballot_w64(h) != 0;
    };
    auto next_surviving = [&](int s) -> int {
      while (s < nsl && !survives(s)) ++s;
      return s;
    };
    float px0, px1, py0, py1, pz0, pz1;
    auto load = [&](int s) {
      const int o = sub_list[s] * NN_SUB + lane;
      px0 = txp[o];
      px1 = txp[o + 64];
      py0 = typ[o];
      py1 = typ[o + 64];
      pz0 = tzp[o];
      pz1 = tzp[o + 64];
    };
    int cur = next_surviving(0);
    if (cur < nsl) load(cur);
    while (cur < nsl) {
      __syncthreads();  // previous sub-tile fully consumed
      stage[0][lane] = px0;
      stage[0][64 + lane] = px1;
      stage[1][lane] = py0;
      stage[1][64 + lane] = py1;
      stage[2][lane] = pz0;
      stage[2][64 + lane] = pz1;
      __syncthreads();
      int nxt = next_surviving(cur + 1);  // judged with the thresholds before this scan
      if (nxt < nsl) load(nxt);           // in flight during the scan
      const int jbase = sub_list[cur] * NN_SUB;
#pragma unroll 2
      for (int g = 0; g < NN_SUB; g += NNF_G) {
        float X[NNF_G], Y[NNF_G], Z[NNF_G];
        const float4 xa = *reinterpret_cast<const float4*>(&stage[0][g]);  // same address in every lane:
        const float4 xb = *reinterpret_cast<const float4*>(&stage[0][g + 4]);  // LDS broadcast
        const float4 ya = *reinterpret_cast<const float4*>(&stage[1][g]);
        const float4 yb = *reinterpret_cast<const float4*>(&stage[1][g + 4]);
        const float4 za = *reinterpret_cast<const float4*>(&stage[2][g]);
        const float4 zb = *reinterpret_cast<const float4*>(&stage[2][g + 4]);
        X[0] = xa.x; X[1] = xa.y; X[2] = xa.z; X[3] = xa.w; X[4] = xb.x; X[5] = xb.y; X[6] = xb.z; X[7] = xb.w;
        Y[0] = ya.x; Y[1] = ya.y; Y[2] = ya.z; Y[3] = ya.w; Y[4] = yb.x; Y[5] = yb.y; Y[6] = yb.z; Y[7] = yb.w;
        Z[0] = za.x; Z[1] = za.y; Z[2] = za.z; Z[3] = za.w; Z[4] = zb.x; Z[5] = zb.y; Z[6] = zb.z; Z[7] = zb.w;
        float e[NNF_G][1], m[1];
        group_estimates<1>(qx, qy, qz, X, Y, Z, e, m);
        if (__builtin_amdgcn_ballot_w64(m[0] <= T[0]) != 0)
          group_exact<1, true>(qx, qy, qz, X, Y, Z, e, jbase + g, tperm, bd, bj, T);
      }
      // the prefetched candidate was judged before this scan tightened the thresholds
      if (nxt < nsl && !survives(nxt)) {
        nxt = next_surviving(nxt + 1);
        if (nxt < nsl) load(nxt);
      }
      cur = nxt;
    }
    __syncthreads();  // lists are rebuilt by the next round
  }

  if (live) best[i] = ((nn_key_t)__float_as_uint(bd[0]) << 32) | (nn_key_t)(unsigned)bj[0];
}

void launch_nn_pruned(const NnArgs& a, const nn_key_t* seed, int seed_scale, const NnBoxes& b, hipStream_t s) {
  const int ntiles = a.nt_pad / NN_TILE;
  hipLaunchKernelGGL(nn_pruned_kernel, dim3((a.nq + 63) / 64), dim3(64), 0, s, a.qx, a.qy, a.qz, a.nq, a.tx, a.ty, a.tz,
                     ntiles, b.ox, b.oy, b.oz, b.tperm, b.qperm, b.tbox, b.tbox_stride, b.sbox, b.sbox_stride, seed,
                     seed_scale, a.best, a.stop);
}

// First-sweep seeds without a brute-force pre-pass: the target whose Morton code is
// nearest to the query's (binary search in the sorted target keys).  Any index is a
// valid seed -- it only sets the initial search radius.
__global__ void seed_morton_kernel(const unsigned* __restrict__ qkeys, int nq, const unsigned* __restrict__ tkeys,
                                   const int* __restrict__ tperm, int nt, nn_key_t* __restrict__ seed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const unsigned k = qkeys[i];
  int lo = 0, hi = nt;  // first position with tkeys[pos] >= k
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (tkeys[mid] < k) lo = mid + 1; else hi = mid;
  }
  int pos = lo < nt ? lo : nt - 1;
  if (pos > 0 && lo < nt && (k - tkeys[pos - 1]) < (tkeys[pos] - k)) pos = pos - 1;
  seed[i] = (nn_key_t)(unsigned)tperm[pos];
}

void launch_seed_morton(const unsigned* qkeys, int nq, const unsigned* tkeys, const int* tperm, int nt, nn_key_t* seed,
                        hipStream_t s) {
  if (nq <= 0) return;
  hipLaunchKernelGGL(seed_morton_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, qkeys, nq, tkeys, tperm, nt, seed);
}

}  // namespace icpk
