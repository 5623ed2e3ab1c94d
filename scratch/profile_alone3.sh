#!/bin/bash
# A second collection of the single-kernel durations (another box of the pool, for the box-to-box range):  gpurun -- scratch/profile_alone3.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round3_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "batch,path,kernel,workgroups,workgroup_size,queue,launches,mean_us,median_us,min_us,max_us" > $O/kernel_alone_64_9_2.csv
for B in 4096 65536; do
  reps=400; slots=36; [ $B = 65536 ] && { reps=60; slots=3; }
  for p in modulate demod_mf demod_zf demod_mf_ic2 demod_zf_ic2; do
    rocprofv3 --kernel-trace --output-format csv -d /tmp/alone/${p}_$B -o t -- python3 $R/scratch/run_kernel.py $p $B $reps $slots > /dev/null 2>&1
    python3 $R/scratch/trace_by_shape.py /tmp/alone/${p}_$B/t_kernel_trace.csv | grep -E "k_row|k_est" | awk -v b=$B -v p=$p -v r=$reps -F'"' '{split($3,a,","); if (a[5]+0 >= r/2) print b "," p "," "\"" $2 "\"" $3}' >> $O/kernel_alone_64_9_2.csv
  done
done
echo "shape_batch,path,kernel,workgroups,workgroup_size,queue,launches,mean_us,median_us,min_us,max_us" > $O/shape_kernel_durations.csv
for b in 8192 65536; do for p in demod_mf demod_mf_ic2 demod_zf_ic2; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/alone/k128_${p}_$b -o t -- python3 $R/scratch/run_kernel.py $p $b 40 2 128 15 4 > /dev/null 2>&1
  python3 $R/scratch/trace_by_shape.py /tmp/alone/k128_${p}_$b/t_kernel_trace.csv | grep -E "k_row_receive" | awk -v b=128_15_4_$b -v p=$p -F'"' '{split($3,a,","); if (a[5]+0 >= 20) print b "," p "," "\"" $2 "\"" $3}' >> $O/shape_kernel_durations.csv
done; done
python3 $R/bench.py --no-paths --no-cpu-baseline --sustained-seconds 0 > $O/bench_headline.json 2>/dev/null
cat $O/kernel_alone_64_9_2.csv $O/shape_kernel_durations.csv | cut -d, -f1-2,9-
