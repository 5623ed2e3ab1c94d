#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3g; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "rc=$?" >> $O/smoke.txt; grep -v amdgpu.ids $O/smoke.txt | tail -8
cd /tmp && export TMPDIR=/tmp
for B in 4096 65536; do
  reps=400; slots=36; [ $B = 65536 ] && { reps=60; slots=3; }
  rocprofv3 --kernel-trace --output-format csv -d /tmp/alone/modulate_$B -o t -- python3 $R/scratch/run_kernel.py modulate $B $reps $slots > /dev/null 2>&1
  python3 $R/scratch/trace_by_shape.py /tmp/alone/modulate_$B/t_kernel_trace.csv | grep -E "k_row" | awk -v b=$B -v p=modulate -v r=$reps -F'"' '{split($3,a,","); if (a[5]+0 >= r/2) print b "," p "," "\"" $2 "\"" $3}'
done | tee $O/modulate_alone.csv
