// tracker_bench.cpp -- the frame path as a C++ caller drives it (SLAM.cpp:229-305 -> icp.cpp:28), timed without an
// interpreter in the way: per frame pair icpk_backproject_pair (one 640x480 uint16 upload, the previous frame is
// resident), icpk_align (16 iterations at most, threshold 1e-4: SLAM.cpp:277) and icpk_get_trace -- the very call
// sequence of bench.py's tracker_path (which makes it through ctypes) and of icp::Tracker::getTransformation.
// Host-only C++ over the C ABI; bench.py runs it as a child process, tests/test_gpu_cpp_mirror.py checks that its
// transforms are the Python path's bit for bit.
//
//   tracker_bench <frames.u16> <rows> <cols> <n_frames> <rounds> <filter 0|1> <resident 0|1> [device [registered 0|1]]
// registered = 1: the frame buffer is pinned with icpk_register_host_buffer first (a caller with long-lived frame buffers)
// frames.u16: n_frames images of rows x cols uint16, back to back.  Prints one JSON line.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "icpk.h"

static uint32_t crc32_update(uint32_t crc, const void* data, size_t n) {  // (zlib's polynomial: comparable with zlib.crc32)
  static uint32_t table[256];
  if (!table[1])
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
  crc = ~crc;
  const unsigned char* p = static_cast<const unsigned char*>(data);
  for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
  return ~crc;
}

int main(int argc, char** argv) {
  if (argc < 8) {
    std::fprintf(stderr, "usage: %s frames.u16 rows cols n_frames rounds filter resident [device]\n", argv[0]);
    return 2;
  }
  const int rows = std::atoi(argv[2]), cols = std::atoi(argv[3]), nf = std::atoi(argv[4]), rounds = std::atoi(argv[5]);
  const int filter = std::atoi(argv[6]), resident = std::atoi(argv[7]);
  const int device = argc > 8 ? std::atoi(argv[8]) : 0;
  const int registered = argc > 9 ? std::atoi(argv[9]) : 0;
  if (rows <= 0 || cols <= 0 || nf < 2 || rounds < 1) return 2;
  const size_t px = (size_t)rows * cols;
  std::vector<uint16_t> frames(px * nf);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(frames.data(), sizeof(uint16_t), frames.size(), f) != frames.size()) {
    std::fprintf(stderr, "cannot read %zu pixels from %s\n", frames.size(), argv[1]);
    return 2;
  }
  std::fclose(f);

  icpk_ctx* ctx = nullptr;
  int rc = icpk_create(&ctx, device);
  if (rc != ICPK_OK) {
    std::fprintf(stderr, "icpk_create: %d\n", rc);
    return 1;
  }
  if (registered && (rc = icpk_register_host_buffer(ctx, frames.data(), frames.size() * sizeof(uint16_t))) != ICPK_OK) {
    std::fprintf(stderr, "icpk_register_host_buffer: %d (%s)\n", rc, icpk_last_error(ctx));
    icpk_destroy(ctx);
    return 1;
  }
  const float camR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, camP[3] = {5, 5, 5};  // icp.cpp:49, 53
  icpk_params par;
  icpk_default_params(&par);
  par.max_iterations = 16;
  par.threshold = 1e-4f;
  std::vector<float> trR(17 * 9), trT(17 * 3), trM(17);
  std::vector<int32_t> trP(17);
  uint32_t crc = 0;
  long its = 0;
  double t_bp = 0, t_al = 0, t_tr = 0;
  using clk = std::chrono::steady_clock;
  auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  auto pair = [&](int i, bool first, bool timed) -> int {
    const uint16_t* cur = frames.data() + px * i;
    const uint16_t* prev = (first || !resident) ? frames.data() + px * (i - 1) : nullptr;
    float T[16];
    icpk_stats st;
    int32_t ns = 0, nt = 0, niter = 0;
    const auto t0 = clk::now();
    int r = icpk_backproject_pair(ctx, cur, prev, rows, cols, ICPK_FX, ICPK_CX, nullptr, camR, camP, filter, 25000, 1000,
                                  1, -1, -1, &ns, &nt);
    if (r != ICPK_OK) return r;
    const auto t1 = clk::now();
    r = icpk_align(ctx, &par, T, &st);
    if (r < 0) return r;
    const auto t2 = clk::now();
    icpk_get_trace(ctx, &niter, trR.data(), trT.data(), trP.data(), trM.data());
    const auto t3 = clk::now();
    if (timed) {
      t_bp += secs(t0, t1);
      t_al += secs(t1, t2);
      t_tr += secs(t2, t3);
      its += st.iterations;
      crc = crc32_update(crc, T, sizeof(T));
    }
    return ICPK_OK;
  };
  for (int i = 1; i < nf; ++i)  // warm-up: one pass over the sequence
    if ((rc = pair(i, i == 1, false)) != ICPK_OK) break;
  int n = 0;
  const auto t0 = clk::now();
  for (int r = 0; r < rounds && rc == ICPK_OK; ++r)
    for (int i = 1; i < nf; ++i) {  // (the sequence wraps around: frame 1 follows the last frame only through an explicit `previous`)
      if ((rc = pair(i, i == 1, true)) != ICPK_OK) break;
      ++n;
    }
  const double dt = secs(t0, clk::now());
  if (rc != ICPK_OK) {
    std::fprintf(stderr, "failed: %d (%s)\n", rc, icpk_last_error(ctx));
    icpk_destroy(ctx);
    return 1;
  }
  std::printf("{\"frame_pairs_per_s\": %.3f, \"ms_per_pair\": %.6f, \"mean_iterations\": %.3f, \"pairs\": %d, "
              "\"points\": [%d, %d], \"host_ms_per_call\": {\"backproject_pair\": %.4f, \"align\": %.4f, \"get_trace\": %.4f}, "
              "\"transforms_crc32\": \"%08x\"}\n",
              n / dt, dt / n * 1e3, (double)its / n, n, icpk_source_size(ctx), icpk_target_size(ctx), t_bp / n * 1e3,
              t_al / n * 1e3, t_tr / n * 1e3, crc);
  icpk_destroy(ctx);
  return 0;
}
