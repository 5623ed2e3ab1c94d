#!/bin/bash
# A/B on ONE box: tree vs the commit before (scratch/ab/head), IC kernels on the vector ALU (their default at K <= 64), rocprofv3 kernel durations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3y; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu -k "advanced or golden or ic or matrix_cores or phase" > $O/sel.txt 2>&1; echo "rc=$?" >> $O/sel.txt; tail -4 $O/sel.txt
cd /tmp && export TMPDIR=/tmp
run() {   # tag path B reps slots K M L
  rocprofv3 --kernel-trace --output-format csv -d /tmp/aby/$1_$2_$3_$6 -o t -- python3 $R/scratch/run_kernel.py $2 $3 $4 $5 $6 $7 $8 > /dev/null 2>&1
  python3 $R/scratch/trace_by_shape.py /tmp/aby/$1_$2_$3_$6/t_kernel_trace.csv | grep "k_row_receive" | awk -F'"' -v t=$1 -v p=$2 -v b=$3 '{split($3,a,","); if (a[5]+0 >= 20) printf "%-6s %-13s B=%-6s %-40s n=%s mean %s median %s min %s\n", t, p, b, $2, a[5], a[6], a[7], a[8]}'
  rm -rf /tmp/aby/$1_$2_$3_$6
}
for round in 1 2; do
  for v in tree head; do
    if [ $v = head ]; then export GFDM_HIP_LIB=$R/scratch/ab/head/libgfdm_hip.so; else unset GFDM_HIP_LIB; fi
    run $v demod_mf_ic2 4096 400 36 64 9 2
    run $v demod_zf_ic2 4096 400 36 64 9 2
    run $v demod_mf_ic2 65536 40 3 64 9 2
    run $v demod_zf_ic2 65536 40 3 64 9 2
    run $v demod_zf_ic2 65536 40 3 32 5 2
    GFDM_MX=0 run $v demod_mf_ic2 8192 200 12 128 15 4
  done
done | tee $O/ab_icgrp.txt
