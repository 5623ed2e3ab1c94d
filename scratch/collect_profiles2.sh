#!/bin/bash
# gpurun_out/round2 (scratch/profile_round2.sh) -> profiles/<round>/    usage: scratch/collect_profiles2.sh r02
set -e
D=profiles/${1:-r02}; O=gpurun_out/round2
mkdir -p $D
cp $O/bench_default_plain.json $O/bench_default_under_rocprofv3.json $O/bench_cfg3.json $O/bench_cfg4.json $O/bench_cfg5.json $D/
cp $O/bench_default_kernel_stats.csv $O/bench_default_domain_stats.csv $O/bench_default_kernel_durations_by_launch_shape.csv $D/
cp $O/bench_est_kernel_stats.csv $O/bench_est_kernel_durations_by_launch_shape.csv $D/
cp $O/pmc_hbm_traffic_summary.csv $O/pmc_sq_counters_summary.csv $O/shape_kernel_durations.csv $D/
for f in bench_est bench_tx bench_frames bw_probe shape_64_9_2_65536 shape_32_5_2_65536 shape_128_15_4_8192 shape_128_15_4_65536 shape_256_31_2_8192 shape_128_21_2_4096 shape_16_7_2_65536_jit shape_96_25_2_4096 shape_16_127_2_4096_generic shape_1024_15_2_2048_jit bench_stages; do grep -v "amdgpu.ids" $O/$f.txt > $D/$f.txt; done
ls -la $D
