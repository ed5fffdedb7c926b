// tests/cpp/fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY: a stand-in for librccl that lets two (or more)
// processes SHARING ONE GPU run the world > 1 paths of csrc/icpk_comm.cpp.  Real RCCL refuses two
// ranks on one device, so on the 1-GPU test box the rank-dependent logic behind the C ABI
// (partition, staging, row re-ordering of icpk_comm_gather_results, the non-root side of
// icpk_comm_broadcast_target, icpk_comm_allreduce_sums) could otherwise only run with world = 1.
// Loaded through ICPK_RCCL_LIB (a test hook of icpk_comm.cpp); never shipped, never used by bench.py.
//
// Implements the nine symbols icpk_comm.cpp resolves, with the semantics it relies on, over a
// POSIX shared-memory segment: every collective = stream sync, device->shm, barrier, shm->device,
// barrier.  Slow and simple by design.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and enums only
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
constexpr size_t DATA_BYTES = 64u << 20;
struct Shm {
  std::atomic<int> joined;
  std::atomic<int> arrive;
  std::atomic<int> phase;
  char pad[64 - 3 * sizeof(int)];
  unsigned char data[DATA_BYTES];
};
}  // namespace

struct ncclComm {
  Shm* shm;
  int rank, world;
  char name[96];
};

namespace {
void barrier(ncclComm* c) {
  const int ph = c->shm->phase.load();
  if (c->shm->arrive.fetch_add(1) + 1 == c->world) {
    c->shm->arrive.store(0);
    c->shm->phase.store(ph + 1);
  } else {
    while (c->shm->phase.load() == ph) usleep(50);
  }
}
size_t dsize(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id, 0, sizeof(*id));
  std::snprintf(id->internal, sizeof(id->internal), "/icpk_fake_rccl_%d_%ld", (int)getpid(), (long)random());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  ncclComm* c = new ncclComm();
  c->rank = rank;
  c->world = nranks;
  std::snprintf(c->name, sizeof(c->name), "%s", id.internal);
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(Shm)) != 0) return ncclSystemError;
  c->shm = static_cast<Shm*>(mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
  close(fd);
  if (c->shm == MAP_FAILED) return ncclSystemError;
  c->shm->joined.fetch_add(1);  // (a fresh segment is zero-filled)
  while (c->shm->joined.load() < nranks) usleep(100);
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  barrier(c);
  if (c->rank == 0) shm_unlink(c->name);
  munmap(c->shm, sizeof(Shm));
  delete c;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake rccl error"; }
ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t c,
                           hipStream_t s) {
  const size_t bytes = count * dsize(t);
  if (bytes > DATA_BYTES) return ncclInvalidArgument;
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  if (c->rank == root && hipMemcpy(c->shm->data, send, bytes, hipMemcpyDeviceToHost) != hipSuccess)
    return ncclUnhandledCudaError;
  barrier(c);
  if (hipMemcpy(recv, c->shm->data, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  barrier(c);
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t t, ncclComm_t c,
                           hipStream_t s) {
  const size_t bytes = sendcount * dsize(t);
  if (bytes * c->world > DATA_BYTES) return ncclInvalidArgument;
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->shm->data + bytes * c->rank, send, bytes, hipMemcpyDeviceToHost) != hipSuccess)
    return ncclUnhandledCudaError;
  barrier(c);
  if (hipMemcpy(recv, c->shm->data, bytes * c->world, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  barrier(c);
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c,
                           hipStream_t s) {
  if (t != ncclFloat64 || op != ncclSum || count * 8 * c->world > DATA_BYTES) return ncclInvalidArgument;
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  double* all = reinterpret_cast<double*>(c->shm->data);
  if (hipMemcpy(all + count * c->rank, send, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  barrier(c);
  double* acc = static_cast<double*>(std::malloc(count * 8));
  for (size_t k = 0; k < count; ++k) {
    double v = 0.0;
    for (int r = 0; r < c->world; ++r) v += all[count * r + k];  // rank order
    acc[k] = v;
  }
  const bool ok = hipMemcpy(recv, acc, count * 8, hipMemcpyHostToDevice) == hipSuccess;
  std::free(acc);
  barrier(c);
  return ok ? ncclSuccess : ncclUnhandledCudaError;
}

}  // extern "C"
