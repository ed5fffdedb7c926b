"""icp_slam_prototype_amd -- MI355X-native ICP inner loop.

The compute path lives in lib/libicpk.so (hand-written HIP for gfx950 behind
the C ABI of include/icpk.h).  This package is the thin Python binding used by
tests/ and bench.py; `synth` generates the benchmark workloads.
"""
