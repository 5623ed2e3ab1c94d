#!/bin/bash
# rocprofv3 kernel durations of every K=64 M=9 path, ONE kernel on the GPU at a time (scratch/run_kernel.py over a ring of buffers):
#   gpurun -- scratch/profile_alone.sh   ->  gpurun_out/round2/kernel_alone_64_9_2.csv
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round2; T=/tmp/alone; mkdir -p $O $T
cd /tmp && export TMPDIR=/tmp
echo "batch,path,kernel,workgroups,workgroup_size,queue,launches,mean_us,median_us,min_us,max_us" > $O/kernel_alone_64_9_2.csv
for B in 4096 65536; do
  reps=400; slots=36; [ $B = 65536 ] && { reps=60; slots=3; }
  for p in modulate demod_mf demod_zf demod_mf_ic2 demod_zf_ic2 frames_zf_ic2_est estimate_frame; do
    rocprofv3 --kernel-trace --output-format csv -d $T/${p}_$B -o t -- python3 $R/scratch/run_kernel.py $p $B $reps $slots > /dev/null 2>&1
    python3 $R/scratch/trace_by_shape.py $T/${p}_$B/t_kernel_trace.csv | grep -E "k_row|k_est" | awk -v b=$B -v p=$p -v r=$reps -F'"' '{split($3,a,","); if (a[5]+0 >= r/2) print b "," p "," "\"" $2 "\"" $3}' >> $O/kernel_alone_64_9_2.csv
  done
done
cat $O/kernel_alone_64_9_2.csv
