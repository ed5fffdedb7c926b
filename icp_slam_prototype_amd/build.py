"""Build recipe for lib/libicpk.so (hipcc, gfx950 only) -- run by
__graft_entry__.build().  hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libicpk.so")

SOURCES = ["icpk_api.cpp", "icpk_comm.cpp", "kernels_nn.hip", "kernels_reduce.hip", "kernels_transform.hip",
           "kernels_backproject.hip", "kernels_sort.hip", "kernels_nn_pruned.hip", "kernels_loop.hip", "kernels_grid.hip", "kernels_frontend.hip"]

# -ffp-contract=off: the exact kernels spell out every fma they want; nothing may
# be fused behind their back (host solve included).  No -ffast-math anywhere.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-result", "-I", os.path.join(ROOT, "include"), "-I", CSRC]


def _newest_src():
    paths = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    paths.append(os.path.join(ROOT, "include", "icpk.h"))
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=False, extra=(), out=None):
    """out: alternative output path (diagnostic builds, e.g. tools/stamp_pruned.py)."""
    os.makedirs(LIBDIR, exist_ok=True)
    if out is None and not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_src():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for s in SOURCES:
        o = os.path.join(LIBDIR if out is None else os.path.dirname(out), s.rsplit(".", 1)[0] + ".o")
        cmd = [hipcc] + FLAGS + list(extra) + ["-x", "hip", "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        objs.append(o)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out or LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out or LIB


CPP_TEST = os.path.join(LIBDIR, "test_icp_align")


def build_cpp_test(force=False):
    """Host-only C++ program over icp_align.hpp + the C ABI (g++, links -licpk)."""
    src = os.path.join(ROOT, "tests", "cpp", "test_icp_align.cpp")
    hdr = os.path.join(HERE, "include", "icp_align.hpp")
    build()
    newest = max(os.path.getmtime(p) for p in (src, hdr, LIB))
    if not force and os.path.exists(CPP_TEST) and os.path.getmtime(CPP_TEST) >= newest:
        return CPP_TEST
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), "-I",
                           os.path.join(HERE, "include"), src, "-L", LIBDIR, "-licpk", "-Wl,-rpath,$ORIGIN",
                           "-Wl,-rpath-link,/opt/rocm/lib", "-o", CPP_TEST])
    return CPP_TEST


TRACKER_BENCH = os.path.join(LIBDIR, "tracker_bench")


def build_tracker_bench(force=False):
    """Host-only C++ program over the C ABI: the frame path timed without an interpreter (bench.py's tracker_path runs
    it as a child process)."""
    src = os.path.join(ROOT, "tests", "cpp", "tracker_bench.cpp")
    build()
    newest = max(os.path.getmtime(p) for p in (src, LIB, os.path.join(ROOT, "include", "icpk.h")))
    if not force and os.path.exists(TRACKER_BENCH) and os.path.getmtime(TRACKER_BENCH) >= newest:
        return TRACKER_BENCH
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-L", LIBDIR,
                           "-licpk", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link,/opt/rocm/lib", "-o", TRACKER_BENCH])
    return TRACKER_BENCH


FAKE_RCCL = os.path.join(LIBDIR, "libfake_rccl.so")


def build_fake_rccl(force=False):
    """TEST INFRASTRUCTURE: tests/cpp/fake_rccl.cpp (collectives over shared memory for ranks that share
    one GPU), loaded through ICPK_RCCL_LIB by tests/test_gpu_comm_two_ranks.py only."""
    src = os.path.join(ROOT, "tests", "cpp", "fake_rccl.cpp")
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and os.path.exists(FAKE_RCCL) and os.path.getmtime(FAKE_RCCL) >= os.path.getmtime(src):
        return FAKE_RCCL
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "--offload-arch=gfx950", src, "-o",
                           FAKE_RCCL, "-lrt"])
    return FAKE_RCCL


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_cpp_test(force="--force" in sys.argv))
    print(build_tracker_bench(force="--force" in sys.argv))
