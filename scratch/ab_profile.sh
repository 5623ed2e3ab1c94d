#!/bin/bash
# A/B of two builds of libgfdm_hip.so under rocprofv3 on the same box: ab_profile.sh <tag> <lib path or ""> <path> ...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
tag=$1; lib=$2; shift 2
[ -n "$lib" ] && export GFDM_HIP_LIB=$lib GFDM_PKG=$R/scratch/old_pkg
for p in "$@"; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ab_$tag -o $p -- python3 $R/scratch/run_kernel.py $p 4096 400 36 > /dev/null 2>&1
  python3 $R/scratch/trace_by_shape.py $R/gpurun_out/ab_$tag/${p}_kernel_trace.csv | grep "k_row" | grep -v "modulate.*,1,1\b" | sed "s/^/$tag $p /"
done
