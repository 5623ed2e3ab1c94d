#!/bin/bash
# gpurun_out/round3 (scratch/profile_round3.sh) -> profiles/r03/    usage: scratch/collect_profiles3.sh
set -e
D=profiles/r03; O=gpurun_out/round3
mkdir -p $D
cp $O/bench_default_plain.json $O/bench_default_under_rocprofv3.json $O/bench_cfg3.json $O/bench_cfg4.json $O/bench_cfg5.json $D/
cp $O/bench_default_kernel_stats.csv $O/bench_default_domain_stats.csv $O/bench_default_kernel_durations_by_launch_shape.csv $D/
cp $O/kernel_alone_64_9_2.csv $O/pmc_hbm_traffic_summary.csv $O/pmc_sq_counters_summary.csv $O/shape_kernel_durations.csv $D/
for f in shape_32_5_2_65536 shape_128_21_2_4096 shape_16_127_2_4096_generic; do grep -v "amdgpu.ids" $O/$f.txt > $D/$f.txt; done
ls -la $D
