#!/bin/bash
# Everything profiles/r03/ is assembled from, in one gpurun call:  gpurun --timeout 2700 -- scratch/profile_round3.sh
# (then scratch/collect_profiles3.sh here).  Counters are collected in runs of their own (--pmc only), the program directly after "--".
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round3
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the default bench, plain and under rocprofv3 (kernel trace + stats), and the other configurations
python3 $R/bench.py > $O/bench_default_plain.json 2> $O/bench_default_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 $R/bench.py --no-cpu-baseline > $O/bench_default_under_rocprofv3.json 2> $O/bench_under.err
for c in cfg3 cfg4 cfg5; do python3 $R/bench.py --config $c --cpu-seconds 3 > $O/bench_$c.json 2> $O/bench_$c.err; done
# 2. kernel durations, ONE kernel on the GPU at a time: K=64 M=9 every path at 4096 / 65536; cfg4 / cfg5 shapes at 8192 / 65536
echo "batch,path,kernel,workgroups,workgroup_size,queue,launches,mean_us,median_us,min_us,max_us" > $O/kernel_alone_64_9_2.csv
for B in 4096 65536; do
  reps=400; slots=36; [ $B = 65536 ] && { reps=60; slots=3; }
  for p in modulate demod_mf demod_zf demod_mf_ic2 demod_zf_ic2 frames_zf_ic2_est estimate_frame; do
    rocprofv3 --kernel-trace --output-format csv -d /tmp/alone/${p}_$B -o t -- python3 $R/scratch/run_kernel.py $p $B $reps $slots > /dev/null 2>&1
    python3 $R/scratch/trace_by_shape.py /tmp/alone/${p}_$B/t_kernel_trace.csv | grep -E "k_row|k_est" | awk -v b=$B -v p=$p -v r=$reps -F'"' '{split($3,a,","); if (a[5]+0 >= r/2) print b "," p "," "\"" $2 "\"" $3}' >> $O/kernel_alone_64_9_2.csv
  done
done
# ... and the matrix-core form of the IC rounds forced on K=64 (not the default there), for the record
for B in 4096 65536; do
  reps=400; slots=36; [ $B = 65536 ] && { reps=60; slots=3; }
  for p in demod_mf_ic2 demod_zf_ic2; do
    GFDM_MX=2 rocprofv3 --kernel-trace --output-format csv -d /tmp/alone/mx_${p}_$B -o t -- python3 $R/scratch/run_kernel.py $p $B $reps $slots > /dev/null 2>&1
    python3 $R/scratch/trace_by_shape.py /tmp/alone/mx_${p}_$B/t_kernel_trace.csv | grep -E "k_row_receive" | awk -v b=$B -v p=${p}_mx -v r=$reps -F'"' '{split($3,a,","); if (a[5]+0 >= r/2) print b "," p "," "\"" $2 "\"" $3}' >> $O/kernel_alone_64_9_2.csv
  done
done
for shape in "128 15 4" "256 31 2"; do for b in 8192 65536; do
  tag=$(echo $shape | tr ' ' '_')_$b
  for p in modulate demod_mf demod_zf demod_mf_ic2 demod_zf_ic2; do
    rocprofv3 --kernel-trace --output-format csv -d $O/shape_trace/$tag/$p -o t -- python3 $R/scratch/run_kernel.py $p $b 40 2 $shape > /dev/null 2>&1
  done
  if [ "$shape" = "128 15 4" ]; then   # the vector-ALU form of the IC rounds beside the matrix-core default
    for p in demod_mf_ic2 demod_zf_ic2; do
      GFDM_MX=0 rocprofv3 --kernel-trace --output-format csv -d $O/shape_trace/$tag/${p}_valu -o t -- python3 $R/scratch/run_kernel.py $p $b 40 2 $shape > /dev/null 2>&1
    done
  fi
done; done
for d in $O/shape_trace/*/*; do python3 $R/scratch/trace_by_shape.py $d/t_kernel_trace.csv | awk -v t=$(basename $(dirname $d)) -v p=$(basename $d) 'NR==1 && !h {print "shape_batch,path," $0; h=1} NR>1 {print t "," p "," $0}'; done | awk 'NR==1 || !/^shape_batch/' > $O/shape_kernel_durations_all.csv
python3 - "$O" <<PYEOF
import csv, sys
O = sys.argv[1]
rows = list(csv.reader(open(O + "/shape_kernel_durations_all.csv")))
keep = [r for r in rows[1:] if ("k_row_modulate" in r[2] if r[1] == "modulate" else "k_row_receive" in r[2])]
w = csv.writer(open(O + "/shape_kernel_durations.csv", "w", newline=""))
w.writerow(rows[0]); w.writerows(keep)
PYEOF
# 3. HBM traffic (FETCH_SIZE and WRITE_SIZE in separate passes) per launch, run name = <path>_<K>_<M>_<L>_<batch>
pmc_hbm() {  # path batch reps slots K M L
  run=$1_$5_$6_$7_$2
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_hbm/$run/$c -o pmc -- python3 $R/scratch/run_kernel.py $1 $2 $3 $4 $5 $6 $7 > $O/pmc_hbm_$run.log 2>&1
  done
}
for p in modulate demod_mf demod_zf demod_mf_ic2 demod_zf_ic2; do pmc_hbm $p 4096 40 40 64 9 2; done
pmc_hbm demod_zf_ic2 65536 12 3 64 9 2
for b in 8192 65536; do pmc_hbm demod_mf_ic2 $b 10 2 128 15 4; done
pmc_hbm demod_zf 8192 10 2 256 31 2
python3 $R/scratch/pmc_summary.py $O/pmc_hbm > $O/pmc_hbm_traffic_summary.csv 2>&1
# 4. SQ / LDS / MFMA counters of the final kernels (8 SQ slots per pass)
pmc_sq() {  # path batch reps slots K M L
  run=$1_$5_$6_$7_$2
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq/$run/a -o pmc -- python3 $R/scratch/run_kernel.py $1 $2 $3 $4 $5 $6 $7 > $O/pmc_sq_$run.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_sq/$run/b -o pmc -- python3 $R/scratch/run_kernel.py $1 $2 $3 $4 $5 $6 $7 >> $O/pmc_sq_$run.log 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_sq/$run/c -o pmc -- python3 $R/scratch/run_kernel.py $1 $2 $3 $4 $5 $6 $7 >> $O/pmc_sq_$run.log 2>&1
}
for p in modulate demod_mf demod_mf_ic2 demod_zf_ic2; do pmc_sq $p 4096 40 40 64 9 2; done
pmc_sq demod_zf_ic2 65536 12 3 64 9 2
pmc_sq demod_mf 8192 10 2 128 15 4; pmc_sq demod_mf_ic2 8192 10 2 128 15 4; pmc_sq demod_zf 8192 10 2 256 31 2
GFDM_MX=0 pmc_sq demod_mf_ic2 8192 10 2 128 15 4 && mv $O/pmc_sq/demod_mf_ic2_128_15_4_8192 $O/pmc_sq/demod_mf_ic2_valu_128_15_4_8192 && pmc_sq demod_mf_ic2 8192 10 2 128 15 4
python3 $R/scratch/pmc_summary.py $O/pmc_sq > $O/pmc_sq_counters_summary.csv 2>&1
# 5. event-timed per-shape tables
python3 $R/scratch/bench_shape.py 32 5 2 65536 0.5 > $O/shape_32_5_2_65536.txt 2>&1
python3 $R/scratch/bench_shape.py 128 21 2 4096 0.35 > $O/shape_128_21_2_4096.txt 2>&1
python3 $R/scratch/bench_shape.py 16 127 2 4096 0.5 > $O/shape_16_127_2_4096_generic.txt 2>&1
# keep the summaries, drop the raw profiler output (gpurun copies back at most 64 MiB)
python3 $R/scratch/trace_by_shape.py $O/bench/bench_kernel_trace.csv | grep -E "^kernel|k_row|k_generic|k_est" > $O/bench_default_kernel_durations_by_launch_shape.csv
cp $O/bench/bench_kernel_stats.csv $O/bench_default_kernel_stats.csv; cp $O/bench/bench_domain_stats.csv $O/bench_default_domain_stats.csv
rm -rf $O/pmc_hbm $O/pmc_sq $O/shape_trace $O/bench $O/*.log
du -sh $O; ls $O | head -80; cat $O/bench_default_plain.json | head -c 1500
