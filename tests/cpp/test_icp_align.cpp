// tests/cpp/test_icp_align.cpp -- exercises the C++ host mirror (icp_align.hpp) the
// way SLAM.cpp would: a sequence of depth frames through icp::Tracker, plus one
// icp::align call on explicit clouds.  Input/output are raw binary files so the
// Python parity test can compare against the oracle.
//
//   test_icp_align <in.bin> <out.bin>
// in : int32 rows, cols, nframes, max_iter; float threshold; uint16 depth[nframes][rows*cols]
// out: per frame pair i=1..nframes-1: int32 status, iterations; float T[16], camR[9], camP[3]
//      then icp::align on (frame1 cloud, frame0 cloud given as SoA after the depth data):
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "icp_align.hpp"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 3;
  int32_t hdr[4];
  float thr;
  if (std::fread(hdr, 4, 4, f) != 4 || std::fread(&thr, 4, 1, f) != 1) return 4;
  const int rows = hdr[0], cols = hdr[1], nframes = hdr[2], max_iter = hdr[3];
  std::vector<std::vector<uint16_t>> frames(nframes, std::vector<uint16_t>((size_t)rows * cols));
  for (auto& fr : frames)
    if (std::fread(fr.data(), 2, fr.size(), f) != fr.size()) return 5;
  int32_t ns = 0, nt = 0;
  std::vector<float> s, t;
  if (std::fread(&ns, 4, 1, f) == 1 && std::fread(&nt, 4, 1, f) == 1) {
    s.resize((size_t)3 * ns);
    t.resize((size_t)3 * nt);
    if (std::fread(s.data(), 4, s.size(), f) != s.size() || std::fread(t.data(), 4, t.size(), f) != t.size()) return 6;
  }
  std::fclose(f);

  FILE* o = std::fopen(argv[2], "wb");
  if (!o) return 7;
  try {
    icp::Engine eng(0);
    icp::Tracker trk(eng);
    trk.params.fixed_iterations = 0;
    for (int i = 1; i < nframes; ++i) {
      float T[16];
      const int rc = trk.getTransformation(frames[i].data(), frames[i - 1].data(), rows, cols, max_iter, thr, T);
      if (rc < 0) {
        std::fprintf(stderr, "getTransformation failed: %d %s\n", rc, eng.last_error());
        return 8;
      }
      const int32_t head[2] = {rc, trk.lastStats.iterations};
      std::fwrite(head, 4, 2, o);
      std::fwrite(T, 4, 16, o);
      std::fwrite(trk.cameraRotation, 4, 9, o);
      std::fwrite(trk.cameraPosition, 4, 3, o);
      float ex, ey, ez;
      icp::toEulerianAngle(trk.cameraRotation, ex, ey, ez);
      const float e[3] = {ex, ey, ez};
      std::fwrite(e, 4, 3, o);
    }
    if (ns > 0) {
      icp::CloudView src{s.data(), s.data() + ns, s.data() + 2 * (size_t)ns, ns};
      icp::CloudView tgt{t.data(), t.data() + nt, t.data() + 2 * (size_t)nt, nt};
      icp::AlignParams p;
      p.solve = ICPK_SOLVE_KABSCH;
      p.max_iterations = max_iter;
      p.fixed_iterations = 1;
      icp::AlignResult r;
      const int rc = icp::align(eng, src, tgt, p, &r);
      const int32_t head[2] = {rc, r.stats.iterations};
      std::fwrite(head, 4, 2, o);
      std::fwrite(r.T, 4, 16, o);
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 9;
  }
  std::fclose(o);
  return 0;
}
