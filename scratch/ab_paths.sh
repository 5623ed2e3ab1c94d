#!/bin/bash
# A/B of library builds on the BASELINE shapes, rocprofv3 kernel durations, one kernel at a time, two interleaved rounds:
#   gpurun -- scratch/ab_paths.sh "<tag> <tag> ..."      ("tree" = the in-tree build; other tags: scratch/ab/<tag>/libgfdm_hip.so)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_paths; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {   # tag path B reps slots K M L
  rocprofv3 --kernel-trace --output-format csv -d /tmp/abp/$1_$2_$3_$6 -o t -- python3 $R/scratch/run_kernel.py $2 $3 $4 $5 $6 $7 $8 > /dev/null 2>&1
  python3 $R/scratch/trace_by_shape.py /tmp/abp/$1_$2_$3_$6/t_kernel_trace.csv | grep -E "k_row_receive|k_row_modulate" | awk -F'"' -v t=$1 -v p=$2 -v b=$3 -v want=$4 '{split($3,a,","); if (a[5]+0 >= want/2) printf "%-6s %-13s B=%-6s %-40s n=%s mean %s median %s min %s\n", t, p, b, $2, a[5], a[6], a[7], a[8]}'
}
for round in 1 2; do
  for tag in $1; do
    if [ "$tag" = "tree" ]; then unset GFDM_HIP_LIB; else export GFDM_HIP_LIB=$R/scratch/ab/$tag/libgfdm_hip.so; fi
    for p in modulate demod_mf demod_zf_ic2; do run $tag $p 4096 400 36 64 9 2; done
    for p in demod_mf demod_zf_ic2; do run $tag $p 65536 40 3 64 9 2; done
    for p in modulate demod_mf demod_mf_ic2; do run $tag $p 8192 200 12 128 15 4; done
    run $tag demod_mf_ic2 65536 30 2 128 15 4
    run $tag demod_zf 8192 100 6 256 31 2
  done
done | tee $O/ab_paths.txt
