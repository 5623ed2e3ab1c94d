#!/bin/bash
# A/B of variant builds (scratch/build_variant.sh) under rocprofv3 --kernel-trace on ONE box, two rounds each, interleaved:
#   gpurun -- scratch/ab_run.sh "<tag> <tag> ..." "<path> <path> ..." [batch]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
tags=$1; paths=$2; B=${3:-4096}
reps=400; slots=36
[ $B -gt 8192 ] && { reps=40; slots=4; }
for round in 1 2; do
  for tag in $tags; do
    export GFDM_HIP_LIB=$R/scratch/ab/$tag/libgfdm_hip.so
    for p in $paths; do
      rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ab/${tag}_${B}_$round -o $p -- python3 $R/scratch/run_kernel.py $p $B $reps $slots > /dev/null 2>&1
      python3 $R/scratch/trace_by_shape.py $R/gpurun_out/ab/${tag}_${B}_$round/${p}_kernel_trace.csv | grep "k_row" | awk -F, -v t=$tag -v p=$p -v r=$round -v want=$reps '$5>=want/2 {printf "%-8s r%s %-14s %-42s n=%s mean %s median %s min %s\n", t, r, p, $1, $5, $6, $7, $8}'
    done
  done
done
