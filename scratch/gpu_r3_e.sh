#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3e; mkdir -p $O
cd $R
GFDM_HIP_LIB=$R/scratch/ab/persist/libgfdm_hip.so timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "128 or 256 or full_size or matrix_cores" > $O/persist_tests.txt 2>&1; echo "rc=$?" >> $O/persist_tests.txt; tail -6 $O/persist_tests.txt
scratch/ab_k128.sh "tree persist"
cd /tmp && export TMPDIR=/tmp
for tag in tree persist; do
  if [ "$tag" = "tree" ]; then unset GFDM_HIP_LIB; else export GFDM_HIP_LIB=$R/scratch/ab/$tag/libgfdm_hip.so; fi
  for p in demod_mf demod_zf; do
    rocprofv3 --kernel-trace --output-format csv -d /tmp/abq/${tag}_$p -o t -- python3 $R/scratch/run_kernel.py $p 8192 100 6 256 31 2 > /dev/null 2>&1
    python3 $R/scratch/trace_by_shape.py /tmp/abq/${tag}_$p/t_kernel_trace.csv | grep "k_row_receive" | awk -F'"' -v t=$tag -v p=$p '{split($3,a,","); if (a[5]+0 >= 20) printf "%-8s %-13s B=8192 %-40s n=%s mean %s median %s min %s\n", t, p, $2, a[5], a[6], a[7], a[8]}'
    rocprofv3 --kernel-trace --output-format csv -d /tmp/abq/${tag}_${p}_128_65536 -o t -- python3 $R/scratch/run_kernel.py $p 65536 30 2 128 15 4 > /dev/null 2>&1
    python3 $R/scratch/trace_by_shape.py /tmp/abq/${tag}_${p}_128_65536/t_kernel_trace.csv | grep "k_row_receive" | awk -F'"' -v t=$tag -v p=$p '{split($3,a,","); if (a[5]+0 >= 20) printf "%-8s %-13s B=65536 %-40s n=%s mean %s median %s min %s\n", t, p, $2, a[5], a[6], a[7], a[8]}'
  done
done | tee $O/ab_more.txt
