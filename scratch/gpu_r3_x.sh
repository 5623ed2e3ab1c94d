#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3x; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o gen -- python3 $R/scratch/bench_shape.py 16 127 2 4096 0.5 > $O/shape.txt 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/generic_16_127_kernel_stats.csv
python3 $R/scratch/trace_by_shape.py $(find $O/prof -name "*kernel_trace.csv" | head -1) | grep -E "^kernel|k_generic" > $O/generic_16_127_kernel_durations.csv
rm -rf $O/prof
grep -v amdgpu $O/shape.txt; head -12 $O/generic_16_127_kernel_stats.csv | cut -c1-200; cat $O/generic_16_127_kernel_durations.csv | cut -c1-220
