set -e
cd $GRAFT_REPO_ROOT
bash tools/prof_bench.sh r03 > gpurun_out/prof_bench_r03.log 2>&1
bash tools/trace_kernels.sh r03_single --workload kinect640x480_30pct --nn-mode grid > gpurun_out/profiles_new/r03_kernel_trace_summary.txt 2>&1
bash tools/trace_kernels.sh r03_batch8 --batch 8 >> gpurun_out/profiles_new/r03_kernel_trace_summary.txt 2>&1
bash tools/pmc_waves.sh --workload kinect640x480_30pct --nn-mode grid > gpurun_out/profiles_new/r03_wave_cycle_breakdown_raw.txt 2>&1
bash tools/timeline_native.sh > gpurun_out/profiles_new/r03_tracker_native_timeline.txt 2>&1
bash tools/timeline_batch.sh 8 8 > gpurun_out/profiles_new/r03_batch8_timeline.txt 2>&1
python bench.py > gpurun_out/profiles_new/r03_bench.json 2> gpurun_out/bench_r03.err
tail -c 600 gpurun_out/profiles_new/r03_bench.json
