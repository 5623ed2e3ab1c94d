#!/bin/bash
# A/B on ONE box: IC rounds on the matrix cores (default) vs on the vector ALU (GFDM_NO_MX=1), rocprofv3 kernel durations, one kernel on
# the GPU at a time, two interleaved rounds.   gpurun -- scratch/ab_mx.sh [libtag]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_mx; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
[ -n "$1" ] && export GFDM_HIP_LIB=$R/scratch/ab/$1/libgfdm_hip.so
run() {   # tag path B reps slots K M L
  rocprofv3 --kernel-trace --output-format csv -d /tmp/abmx/$1_$2_$3_$6 -o t -- python3 $R/scratch/run_kernel.py $2 $3 $4 $5 $6 $7 $8 > /dev/null 2>&1
  python3 $R/scratch/trace_by_shape.py /tmp/abmx/$1_$2_$3_$6/t_kernel_trace.csv | grep "k_row_receive" | awk -F'"' -v t=$1 -v p=$2 -v b=$3 '{split($3,a,","); if (a[5]+0 >= 20) printf "%-6s %-13s B=%-6s %-40s n=%s mean %s median %s min %s\n", t, p, b, $2, a[5], a[6], a[7], a[8]}'
}
for round in 1 2; do
  for v in mx valu; do
    if [ $v = valu ]; then export GFDM_MX=0; else export GFDM_MX=2; fi
    run $v demod_mf_ic2 4096 400 36 64 9 2
    run $v demod_zf_ic2 4096 400 36 64 9 2
    run $v demod_mf_ic2 65536 40 3 64 9 2
    run $v demod_zf_ic2 65536 40 3 64 9 2
    run $v demod_mf_ic2 8192 200 12 128 15 4
    run $v demod_mf_ic2 65536 30 2 128 15 4
    run $v demod_zf_ic2 8192 200 8 128 15 4
  done
done | tee $O/ab_mx.txt
