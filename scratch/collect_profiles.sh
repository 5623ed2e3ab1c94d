#!/bin/bash
# gpurun_out/round (scratch/profile_round.sh) -> profiles/<round>/    usage: scratch/collect_profiles.sh r01
set -e
D=profiles/${1:-r01}; O=gpurun_out/round
mkdir -p $D
cp $O/bench_default_plain.json $O/bench_default_under_rocprofv3.json $D/
cp $O/bench/bench_kernel_stats.csv $D/bench_default_kernel_stats.csv
cp $O/bench/bench_domain_stats.csv $D/bench_default_domain_stats.csv
python3 scratch/trace_by_shape.py $O/bench/bench_kernel_trace.csv | grep -E "^kernel|k_row|k_generic|k_est" > $D/bench_default_kernel_durations_by_launch_shape.csv
cp $O/est/est_kernel_stats.csv $D/bench_est_kernel_stats.csv
python3 scratch/trace_by_shape.py $O/est/est_kernel_trace.csv | grep -E "^kernel|k_row|k_generic|k_est" > $D/bench_est_kernel_durations_by_launch_shape.csv
grep -v "amdgpu.ids" $O/bench_est.txt > $D/bench_est.txt
for f in shape_64_9_2_65536 shape_32_5_2_65536 shape_128_15_4_65536 shape_256_31_2_8192 bench_tx bench_frames; do grep -v "amdgpu.ids" $O/$f.txt > $D/$f.txt; done
ls -la $D
