// tests/cpp/test_icp_align.cpp -- exercises the C++ host mirror (icp_align.hpp) the
// way SLAM.cpp would: a sequence of depth frames through icp::Tracker, plus one
// icp::align call on explicit clouds.  Input/output are raw binary files so the
// Python parity test can compare against the oracle.
//
//   test_icp_align <in.bin> <out.bin>
// in : int32 rows, cols, nframes, max_iter; float threshold; uint16 depth[nframes][rows*cols]
// out: per frame pair i=1..nframes-1: int32 status, iterations; float T[16], camR[9], camP[3]
//      then icp::align on (frame1 cloud, frame0 cloud given as SoA after the depth data): int32
//      status, iterations; float T[16]; then the same pair three times through icp::alignBatch and
//      through the engine-less icp::align(source, target, params, &result): int32 agree (1/0);
//      then the gathered copy of the batch through icp::Comm with world = 1: int32 agree;
//      then frame 0 through icp::filterDepthImage: uint16[rows*cols];
//      then icp::findGlobalKeyPointAssociations(source as key points, target as map, 0.1):
//      int32 status, n_assoc, n_rejected; int32 pairs[2*n_assoc]; float errors[n_assoc]; int32 rejected[n_rejected]
//      then frames 1 / 0 through a fresh icp::Tracker with filterFrames = true: int32 status, iterations; float T[16]
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
      // frame-batch mode and north_star's engine-less call surface give the same bits
      std::vector<icp::FramePair> pairs(3, icp::FramePair{src, tgt});
      std::vector<icp::AlignResult> br;
      const int brc = icp::alignBatch(eng, pairs, p, &br);
      icp::AlignResult r2;
      const int rc2 = icp::align(src, tgt, p, &r2);
      int32_t agree = brc == rc && rc2 == rc && br.size() == 3;
      for (const auto& b : br)
        agree = agree && std::memcmp(b.T, r.T, sizeof(r.T)) == 0 && b.stats.iterations == r.stats.iterations;
      agree = agree && std::memcmp(r2.T, r.T, sizeof(r.T)) == 0;
      std::fwrite(&agree, 4, 1, o);
      // RCCL behind the C ABI from C++ (world of one): gathered rows == local rows
      unsigned char id[ICPK_COMM_ID_BYTES];
      int32_t cagree = icp::Comm::uniqueId(id) == ICPK_OK;
      if (cagree) {
        icp::Comm comm(eng, id, 0, 1);
        std::vector<icp::AlignResult> all;
        int32_t s0 = -1, c0 = -1;
        comm.partition(3, &s0, &c0);
        cagree = comm.status == ICPK_OK && comm.world() == 1 && s0 == 0 && c0 == 3 &&
                 comm.gatherResults(br, 3, &all) == ICPK_OK && all.size() == 3;
        for (size_t k = 0; cagree && k < 3; ++k)
          cagree = std::memcmp(all[k].T, br[k].T, sizeof(r.T)) == 0 && all[k].stats.iterations == br[k].stats.iterations &&
                   all[k].stats.final_pairs == br[k].stats.final_pairs;
        // the query-sharded device loop with one rank: the engine still holds the pair of icp::align(eng, ...) above
        icp::AlignResult rs;
        cagree = cagree && comm.alignQuerySharded(p, &rs) == rc && std::memcmp(rs.T, r.T, sizeof(r.T)) == 0 &&
                 rs.stats.iterations == r.stats.iterations && rs.stats.final_pairs == r.stats.final_pairs;
      }
      std::fwrite(&cagree, 4, 1, o);
      std::vector<uint16_t> img = frames[0];
      if (icp::filterDepthImage(eng, img.data(), rows, cols) != ICPK_OK) return 10;
      std::fwrite(img.data(), 2, img.size(), o);
      std::vector<float> errors;
      std::vector<std::pair<int32_t, int32_t>> assoc;
      std::vector<int32_t> rejected;
      const int krc = icp::findGlobalKeyPointAssociations(eng, src, tgt, errors, assoc, rejected);
      const int32_t kh[3] = {krc, (int32_t)assoc.size(), (int32_t)rejected.size()};
      std::fwrite(kh, 4, 3, o);
      for (const auto& a : assoc) {
        const int32_t pr[2] = {a.first, a.second};
        std::fwrite(pr, 4, 2, o);
      }
      std::fwrite(errors.data(), 4, errors.size(), o);
      std::fwrite(rejected.data(), 4, rejected.size(), o);
    }
    if (nframes >= 2) {  // a fresh tracker with filterDepthImage inside the call (SLAM.cpp:229 + icp.cpp:38-71)
      icp::Tracker flt(eng);
      flt.params.fixed_iterations = 0;
      flt.filterFrames = true;
      float T[16];
      const int32_t rc = flt.getTransformation(frames[1].data(), frames[0].data(), rows, cols, max_iter, thr, T);
      const int32_t head[2] = {rc, flt.lastStats.iterations};
      std::fwrite(head, 4, 2, o);
      std::fwrite(T, 4, 16, o);
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 9;
  }
  std::fclose(o);
  return 0;
}
