// icpk_comm.cpp -- the collectives of the multi-GPU modes behind the C ABI:
// RCCL over xGMI, one process (or host thread) + one icpk_ctx per GPU, so that a C++ host -- the
// SLAM.cpp:277 drop-in -- reaches the frame-batch mode without Python.
//
//   * frame-batch mode (the path shards over independent frame pairs, icp.cpp:541-563): NO
//     per-iteration collective; one ncclBroadcast of a shared target cloud (key frame) and one
//     ncclAllGather of the results (B x 20 floats) per batch;
//   * query-sharded single pair: one ncclAllReduce of the 19 canonical sums + the pair count per
//     iteration (160 bytes, pure latency).
//
// librccl is NOT a link-time dependency: it is opened on the first icpk_comm_* call
// (dlopen "librccl.so.1": the copy already mapped into the process if there is one -- PyTorch
// ships its own -- else ROCm's), so single-GPU users never load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and enums only

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "icpk.h"
#include "icpk_ctx.h"

using namespace icpk;

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  std::string err;
};

RcclApi* rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // Test hook, honoured ONLY when ICPK_TEST_HOOKS=1 is set as well: ICPK_RCCL_LIB names a library that stands in
    // for RCCL (tests/cpp/fake_rccl.cpp), so that two ranks sharing the one GPU of the test box can exercise the
    // world > 1 paths below (real RCCL refuses that).  Without the switch the variable is ignored: a production
    // process never opens a library because an environment variable says so.
    const char* sw = std::getenv("ICPK_TEST_HOOKS");
    const char* over = (sw && sw[0] == '1' && sw[1] == 0) ? std::getenv("ICPK_RCCL_LIB") : nullptr;
    if (over && *over) {
      api.handle = dlopen(over, RTLD_NOW | RTLD_LOCAL);
    } else {
      for (const char* name : {"librccl.so.1", "librccl.so"}) {
        api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (api.handle) break;
      }
    }
    if (!api.handle) {
      const char* e = dlerror();
      api.err = std::string("dlopen librccl.so.1: ") + (e ? e : "not found");
      return;
    }
    auto sym = [&](const char* n) {
      void* p = dlsym(api.handle, n);
      if (!p && api.err.empty()) api.err = std::string("librccl: missing symbol ") + n;
      return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
  });
  return api.err.empty() ? &api : nullptr;
}

}  // namespace

struct icpk_comm_state {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  void* dev = nullptr;   // device staging (send | recv)
  size_t dev_bytes = 0;
  void* host = nullptr;  // pinned staging of the same size
};

namespace {

#define ICPK_RCCL(ctx, call)                                                                   \
  do {                                                                                         \
    ncclResult_t r__ = (call);                                                                 \
    if (r__ != ncclSuccess) {                                                                  \
      (ctx)->err = std::string(#call) + ": " + (rccl() ? rccl()->GetErrorString(r__) : "?");   \
      return ICPK_E_RCCL;                                                                      \
    }                                                                                          \
  } while (0)

int need_comm(icpk_ctx* ctx) {
  if (!ctx) return ICPK_E_ARG;
  if (!ctx->comm || !ctx->comm->comm) return icpk_host_fail(ctx, ICPK_E_NOT_SET, "icpk_comm_init_rccl has not been called");
  return ICPK_OK;
}

int ensure_staging(icpk_ctx* ctx, size_t bytes) {
  icpk_comm_state* c = ctx->comm;
  if (bytes <= c->dev_bytes) return ICPK_OK;
  if (c->dev) ICPK_HIP(ctx, hipFree(c->dev));
  if (c->host) ICPK_HIP(ctx, hipHostFree(c->host));
  c->dev = c->host = nullptr;
  c->dev_bytes = 0;
  ICPK_HIP(ctx, hipMalloc(&c->dev, bytes));
  ICPK_HIP(ctx, hipHostMalloc(&c->host, bytes, hipHostMallocDefault));
  c->dev_bytes = bytes;
  return ICPK_OK;
}

// block-wise shard of n items over `world` ranks (the same rule as batch.partition)
void shard(int n, int world, int rank, int& start, int& count) {
  const int base = n / world, rem = n % world;
  start = rank * base + (rank < rem ? rank : rem);
  count = base + (rank < rem ? 1 : 0);
}

}  // namespace

// n float64 values on the device, summed over the ranks in place, ENQUEUED on the context's stream: no host copy, no
// host wait (the query-sharded device loop calls this once per iteration between K2 and the loop step)
int icpk_comm_allreduce_device(icpk_ctx* ctx, double* dev, int n) {
  int rc = need_comm(ctx);
  if (rc) return rc;
  ICPK_RCCL(ctx, rccl()->AllReduce(dev, dev, (size_t)n, ncclFloat64, ncclSum, ctx->comm->comm, ctx->stream));
  return ICPK_OK;
}

void icpk_comm_release(icpk_ctx* ctx) {
  if (!ctx || !ctx->comm) return;
  icpk_comm_state* c = ctx->comm;
  if (c->comm && rccl()) (void)rccl()->CommDestroy(c->comm);
  if (c->dev) (void)hipFree(c->dev);
  if (c->host) (void)hipHostFree(c->host);
  delete c;
  ctx->comm = nullptr;
}

extern "C" {

int icpk_comm_unique_id(void* id_out) {
  if (!id_out) return ICPK_E_ARG;
  RcclApi* api = rccl();
  if (!api) return ICPK_E_RCCL;
  static_assert(sizeof(ncclUniqueId) == ICPK_COMM_ID_BYTES, "ICPK_COMM_ID_BYTES must match ncclUniqueId");
  ncclUniqueId id;
  if (api->GetUniqueId(&id) != ncclSuccess) return ICPK_E_RCCL;
  std::memcpy(id_out, &id, sizeof(id));
  return ICPK_OK;
}

int icpk_comm_init_rccl(icpk_ctx* ctx, const void* unique_id, int rank, int world) {
  if (!ctx || !unique_id || world < 1 || rank < 0 || rank >= world) return ICPK_E_ARG;
  RcclApi* api = rccl();
  if (!api) return icpk_host_fail(ctx, ICPK_E_RCCL, "librccl.so.1 could not be loaded (dlopen)");
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  icpk_comm_release(ctx);
  ctx->comm = new icpk_comm_state();
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  ncclResult_t r = api->CommInitRank(&ctx->comm->comm, world, id, rank);
  if (r != ncclSuccess) {
    ctx->err = std::string("ncclCommInitRank: ") + api->GetErrorString(r);
    delete ctx->comm;
    ctx->comm = nullptr;
    return ICPK_E_RCCL;
  }
  ctx->comm->rank = rank;
  ctx->comm->world = world;
  return ICPK_OK;
}

int icpk_comm_destroy(icpk_ctx* ctx) {
  if (!ctx) return ICPK_E_ARG;
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  icpk_comm_release(ctx);
  return ICPK_OK;
}

int icpk_comm_rank(const icpk_ctx* ctx) { return ctx && ctx->comm ? ctx->comm->rank : -1; }
int icpk_comm_world(const icpk_ctx* ctx) { return ctx && ctx->comm ? ctx->comm->world : 0; }

void icpk_comm_partition(int32_t n_items, int world, int rank, int32_t* start, int32_t* count) {
  int s = 0, c = 0;
  if (world >= 1 && rank >= 0 && rank < world && n_items >= 0) shard(n_items, world, rank, s, c);
  if (start) *start = s;
  if (count) *count = c;
}

// key frame: the root's target cloud becomes every rank's target (3 * Nt * 4 bytes over xGMI)
int icpk_comm_broadcast_target(icpk_ctx* ctx, int root) {
  int rc = need_comm(ctx);
  if (rc) return rc;
  icpk_comm_state* c = ctx->comm;
  RcclApi* api = rccl();
  if (root < 0 || root >= c->world) return icpk_host_fail(ctx, ICPK_E_ARG, "bad root");
  if (c->rank == root && !ctx->have_tgt) return icpk_host_fail(ctx, ICPK_E_NOT_SET, "root has no target cloud");
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  rc = ensure_staging(ctx, 64);
  if (rc) return rc;
  int32_t* hn = static_cast<int32_t*>(c->host);
  *hn = c->rank == root ? ctx->tgt.n : 0;
  ICPK_HIP(ctx, hipMemcpyAsync(c->dev, hn, sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  ICPK_RCCL(ctx, api->Broadcast(c->dev, c->dev, 1, ncclInt32, root, c->comm, ctx->stream));
  ICPK_HIP(ctx, hipMemcpyAsync(hn, c->dev, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int n = *hn;
  if (n < 0) return icpk_host_fail(ctx, ICPK_E_RCCL, "broadcast of the target size failed");
  // Every rank makes room first and the ranks AGREE on the outcome (one 4-byte all-reduce) before the plane
  // broadcasts: a rank that could not allocate must not leave the others waiting inside ncclBroadcast.
  int alloc_rc = ICPK_OK;
  if (c->rank != root) alloc_rc = icpk_host_ensure_cloud(ctx, ctx->tgt, n);
  double* hf = static_cast<double*>(c->host);  // (float64 sum: the one all-reduce flavour this layer uses anywhere)
  *hf = alloc_rc == ICPK_OK ? 0.0 : 1.0;
  ICPK_HIP(ctx, hipMemcpyAsync(c->dev, hf, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ICPK_RCCL(ctx, api->AllReduce(c->dev, c->dev, 1, ncclFloat64, ncclSum, c->comm, ctx->stream));
  ICPK_HIP(ctx, hipMemcpyAsync(hf, c->dev, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (alloc_rc) return alloc_rc;
  if (*hf != 0.0) return icpk_host_fail(ctx, ICPK_E_HIP, "another rank could not allocate the broadcast target");
  if (n > 0) {  // plane capacities may differ from rank to rank: one message per plane, grouped
    ICPK_RCCL(ctx, api->GroupStart());
    ICPK_RCCL(ctx, api->Broadcast(ctx->tgt.x(), ctx->tgt.x(), (size_t)n, ncclFloat32, root, c->comm, ctx->stream));
    ICPK_RCCL(ctx, api->Broadcast(ctx->tgt.y(), ctx->tgt.y(), (size_t)n, ncclFloat32, root, c->comm, ctx->stream));
    ICPK_RCCL(ctx, api->Broadcast(ctx->tgt.z(), ctx->tgt.z(), (size_t)n, ncclFloat32, root, c->comm, ctx->stream));
    ICPK_RCCL(ctx, api->GroupEnd());
  }
  if (c->rank != root) {
    rc = icpk_host_target_replaced(ctx);
    if (rc) return rc;
  }
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ICPK_OK;
}

// results of a block-partitioned batch: every rank contributes its n_local rows (T 16 floats +
// iterations, status, pairs, mse) and receives all n_total rows in global pair order
int icpk_comm_gather_results(icpk_ctx* ctx, const float* T_local, const icpk_stats* stats_local, int32_t n_local,
                             int32_t n_total, float* T_all, float* stats_all) {
  int rc = need_comm(ctx);
  if (rc) return rc;
  icpk_comm_state* c = ctx->comm;
  RcclApi* api = rccl();
  int s0 = 0, cnt = 0;
  if (n_total < 0) return icpk_host_fail(ctx, ICPK_E_ARG, "bad n_total");
  shard(n_total, c->world, c->rank, s0, cnt);
  if (n_local != cnt || (n_local > 0 && !T_local) || (n_total > 0 && !T_all))
    return icpk_host_fail(ctx, ICPK_E_ARG, "n_local does not match this rank's block of n_total");
  if (n_total == 0) return ICPK_OK;
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  const int bmax = (n_total + c->world - 1) / c->world;
  const size_t row = 20, send = (size_t)bmax * row, recv = send * c->world;
  rc = ensure_staging(ctx, (send + recv) * sizeof(float));
  if (rc) return rc;
  float* hs = static_cast<float*>(c->host);
  float* hr = hs + send;
  float* ds = static_cast<float*>(c->dev);
  float* dr = ds + send;
  std::memset(hs, 0, send * sizeof(float));
  for (int k = 0; k < n_local; ++k) {
    std::memcpy(hs + row * k, T_local + 16 * (size_t)k, 16 * sizeof(float));
    if (stats_local) {  // the three integers travel as int32 bit patterns in their float slots: exact whatever the cloud size
      const int32_t iv[3] = {stats_local[k].iterations, stats_local[k].status, stats_local[k].final_pairs};
      std::memcpy(hs + row * k + 16, iv, sizeof(iv));
      hs[row * k + 19] = stats_local[k].final_mse;
    }
  }
  ICPK_HIP(ctx, hipMemcpyAsync(ds, hs, send * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  ICPK_RCCL(ctx, api->AllGather(ds, dr, send, ncclFloat32, c->comm, ctx->stream));
  ICPK_HIP(ctx, hipMemcpyAsync(hr, dr, recv * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int r = 0; r < c->world; ++r) {
    int rs = 0, rcnt = 0;
    shard(n_total, c->world, r, rs, rcnt);
    for (int k = 0; k < rcnt; ++k) {
      const float* src = hr + ((size_t)r * bmax + k) * row;
      std::memcpy(T_all + 16 * (size_t)(rs + k), src, 16 * sizeof(float));
      if (stats_all) std::memcpy(stats_all + 4 * (size_t)(rs + k), src + 16, 4 * sizeof(float));
    }
  }
  return ICPK_OK;
}

// query-sharded single pair: global sums = sum over ranks (n doubles + one int64 count)
int icpk_comm_allreduce_sums(icpk_ctx* ctx, double* sums, int32_t n, int64_t* count) {
  int rc = need_comm(ctx);
  if (rc) return rc;
  if (!sums || n < 0 || n > 64) return icpk_host_fail(ctx, ICPK_E_ARG, "bad sums");
  icpk_comm_state* c = ctx->comm;
  RcclApi* api = rccl();
  ICPK_HIP(ctx, hipSetDevice(ctx->device));
  rc = ensure_staging(ctx, (size_t)(n + 1) * sizeof(double));
  if (rc) return rc;
  double* h = static_cast<double*>(c->host);
  std::memcpy(h, sums, (size_t)n * sizeof(double));
  h[n] = count ? (double)*count : 0.0;  // exact below 2^53
  ICPK_HIP(ctx, hipMemcpyAsync(c->dev, h, (size_t)(n + 1) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ICPK_RCCL(ctx, api->AllReduce(c->dev, c->dev, (size_t)n + 1, ncclFloat64, ncclSum, c->comm, ctx->stream));
  ICPK_HIP(ctx, hipMemcpyAsync(h, c->dev, (size_t)(n + 1) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ICPK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::memcpy(sums, h, (size_t)n * sizeof(double));
  if (count) *count = (int64_t)(h[n] + 0.5);
  return ICPK_OK;
}

int icpk_comm_barrier(icpk_ctx* ctx) {
  double z = 0.0;
  return icpk_comm_allreduce_sums(ctx, &z, 1, nullptr);
}

}  // extern "C"
