// Diagnostic (tools only): known byte counts in the access shapes of the grid sweep (K1d), for calibrating what the
// rocprofv3 memory-side counters report on gfx950 -- MI355X_MICROARCH.md (HBM section) calibrates FETCH_SIZE only for wide
// coalesced 16-byte-per-lane streaming reads (it reports half of them) and says to calibrate every other shape oneself.
//   reads : stream16 (coalesced float4 per lane), gather16x8 / gather16x4 (groups of 8 / 4 adjacent lanes read 8 / 4 adjacent
//           float4 at a random aligned position: K1d's candidate loads), lookup4 (one int per lane at a random position, one
//           32-byte sector each: the cell_start look-ups)
//   writes: store16 (coalesced float4), scatter4 / scatter8 (one 4- / 8-byte store per lane into its own random 32-byte sector:
//           the caller-order planes / keys of round 2), scatter32 (two adjacent float4 per lane at a random 32-byte aligned
//           position: round 3's records)
// Every position is visited exactly once (an odd-multiplier bijection of a power-of-two range), the buffer is 1 GiB (beyond the
// 256 MiB Infinity Cache), so "bytes asked for" is also "distinct bytes touched".  Run under rocprofv3 --pmc (tools/pmc_traffic.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); std::exit(1); } } while (0)

__device__ __forceinline__ unsigned perm(unsigned i, unsigned mask) { return (i * 2654435761u + 12345u) & mask; }  // bijection of [0, mask + 1)

__global__ void stream16(const float4* __restrict__ p, size_t n, float* __restrict__ out) {
  float s = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = p[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) out[0] = s;
}
// groups of G adjacent lanes read G adjacent float4 at a random G*16-byte aligned position
template <int G>
__global__ void gather16(const float4* __restrict__ p, unsigned ngroups_mask, float* __restrict__ out) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned g = perm(t / G, ngroups_mask);
  const float4 v = p[(size_t)g * G + (t % G)];
  if (v.x + v.y + v.z + v.w == 123.456f) out[0] = v.x;
}
__global__ void lookup4(const int* __restrict__ p, unsigned nsect_mask, float* __restrict__ out) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  const int v = p[(size_t)perm(t, nsect_mask) * 8 + (t & 7)];  // one int inside its own 32-byte sector
  if (v == 123456789) out[0] = 1.f;
}
__global__ void store16(float4* __restrict__ p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void scatter4(float* __restrict__ p, unsigned nsect_mask) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  p[(size_t)perm(t, nsect_mask) * 8 + (t & 7)] = (float)t;
}
__global__ void scatter8(unsigned long long* __restrict__ p, unsigned nsect_mask) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  p[(size_t)perm(t, nsect_mask) * 4 + (t & 3)] = t;
}
__global__ void scatter32(float4* __restrict__ p, unsigned nsect_mask) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t k = (size_t)perm(t, nsect_mask) * 2;
  p[k] = make_float4(1.f, 2.f, 3.f, (float)t);
  p[k + 1] = make_float4(4.f, 5.f, 6.f, (float)t);
}

int main() {
  const size_t bytes = 1ull << 30;
  void* buf = nullptr;
  float* out = nullptr;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(buf, 0, bytes));
  CHECK(hipDeviceSynchronize());
  const unsigned n16 = (unsigned)(bytes / 16);   // float4 elements: 2^26
  const unsigned nsect = (unsigned)(bytes / 32);  // 32-byte sectors: 2^25
  for (int rep = 0; rep < 2; ++rep) {  // (the second round of dispatches is the one to read: everything is warm)
    hipLaunchKernelGGL(stream16, dim3(8192), dim3(256), 0, 0, (const float4*)buf, (size_t)n16, out);
    hipLaunchKernelGGL(gather16<8>, dim3(n16 / 256), dim3(256), 0, 0, (const float4*)buf, n16 / 8 - 1, out);
    hipLaunchKernelGGL(gather16<4>, dim3(n16 / 256), dim3(256), 0, 0, (const float4*)buf, n16 / 4 - 1, out);
    hipLaunchKernelGGL(lookup4, dim3(nsect / 256), dim3(256), 0, 0, (const int*)buf, nsect - 1, out);
    hipLaunchKernelGGL(store16, dim3(8192), dim3(256), 0, 0, (float4*)buf, (size_t)n16);
    hipLaunchKernelGGL(scatter4, dim3(nsect / 256), dim3(256), 0, 0, (float*)buf, nsect - 1);
    hipLaunchKernelGGL(scatter8, dim3(nsect / 256), dim3(256), 0, 0, (unsigned long long*)buf, nsect - 1);
    hipLaunchKernelGGL(scatter32, dim3(nsect / 256), dim3(256), 0, 0, (float4*)buf, nsect - 1);
    CHECK(hipDeviceSynchronize());
  }
  std::printf("asked_bytes stream16 %zu gather16<8> %zu gather16<4> %zu lookup4 %zu (sectors touched: %zu bytes) store16 %zu scatter4 %zu (sectors: %zu) scatter8 %zu (sectors: %zu) scatter32 %zu\n",
              bytes, bytes, bytes, (size_t)nsect * 4, (size_t)nsect * 32, bytes, (size_t)nsect * 4, (size_t)nsect * 32, (size_t)nsect * 8, (size_t)nsect * 32, bytes);
  return 0;
}
