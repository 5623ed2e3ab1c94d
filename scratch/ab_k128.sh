#!/bin/bash
# A/B of variant builds (scratch/build_variant.sh) on the K=128 M=15 L=4 kernels (BASELINE configs[3]), rocprofv3 kernel durations:
#   gpurun -- scratch/ab_k128.sh "<tag> <tag> ..."      ("tree" = the in-tree build)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_k128; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {   # tag path B reps slots
  rocprofv3 --kernel-trace --output-format csv -d /tmp/abk/$1_$2_$3 -o t -- python3 $R/scratch/run_kernel.py $2 $3 $4 $5 128 15 4 > /dev/null 2>&1
  python3 $R/scratch/trace_by_shape.py /tmp/abk/$1_$2_$3/t_kernel_trace.csv | grep "k_row_receive" | awk -F'"' -v t=$1 -v p=$2 -v b=$3 '{split($3,a,","); if (a[5]+0 >= 20) printf "%-8s %-13s B=%-6s %-40s n=%s mean %s median %s min %s\n", t, p, b, $2, a[5], a[6], a[7], a[8]}'
}
for round in 1 2; do
  for tag in $1; do
    if [ "$tag" = "tree" ]; then unset GFDM_HIP_LIB; else export GFDM_HIP_LIB=$R/scratch/ab/$tag/libgfdm_hip.so; fi
    run $tag demod_mf 8192 200 12
    run $tag demod_mf_ic2 8192 200 12
    run $tag demod_mf_ic2 65536 30 2
    run $tag demod_zf_ic2 8192 200 8
  done
done | tee $O/ab_k128.txt
