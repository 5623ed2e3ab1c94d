#!/bin/bash
# Everything profiles/r02/ is assembled from, in one gpurun call:  gpurun --timeout 2400 -- scratch/profile_round2.sh
# (then scratch/collect_profiles2.sh r02 here).  Counters are collected in runs of their own (--pmc only), the program directly after "--".
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round2
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the default bench, plain and under rocprofv3 (kernel trace + stats), and the other configurations
python3 $R/bench.py > $O/bench_default_plain.json 2> $O/bench_default_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 $R/bench.py --no-cpu-baseline > $O/bench_default_under_rocprofv3.json 2> $O/bench_under.err
for c in cfg3 cfg4 cfg5; do python3 $R/bench.py --config $c --cpu-seconds 4 > $O/bench_$c.json 2> $O/bench_$c.err; done
# 2. HBM traffic (FETCH_SIZE and WRITE_SIZE in separate passes) per launch, run name = <path>_<K>_<M>_<L>_<batch>
pmc_hbm() {  # path batch reps slots K M L
  run=$1_$5_$6_$7_$2
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_hbm/$run/$c -o pmc -- python3 $R/scratch/run_kernel.py $1 $2 $3 $4 $5 $6 $7 > $O/pmc_hbm_$run.log 2>&1
  done
}
for p in modulate demod_mf demod_zf demod_mf_ic2 demod_zf_ic2; do pmc_hbm $p 4096 40 40 64 9 2; pmc_hbm $p 65536 12 3 64 9 2; done
pmc_hbm frames_zf_ic2_est 4096 40 40 64 9 2
pmc_hbm estimate_frame 4096 40 40 64 9 2
for b in 8192 65536; do pmc_hbm demod_mf_ic2 $b 10 2 128 15 4; pmc_hbm demod_zf $b 10 2 256 31 2; done
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --output-format csv -d $O/pmc_hbm/probe/$c -o pmc -- $R/scratch/bw_probe > $O/bw_probe_$c.log 2>&1; done
python3 $R/scratch/pmc_summary.py $O/pmc_hbm > $O/pmc_hbm_traffic_summary.csv 2>&1
$R/scratch/bw_probe > $O/bw_probe.txt 2>&1
# 3. SQ / LDS counters of the final kernels (8 SQ slots per pass)
pmc_sq() {  # path batch reps slots K M L
  run=$1_$5_$6_$7_$2
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq/$run/a -o pmc -- python3 $R/scratch/run_kernel.py $1 $2 $3 $4 $5 $6 $7 > $O/pmc_sq_$run.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_sq/$run/b -o pmc -- python3 $R/scratch/run_kernel.py $1 $2 $3 $4 $5 $6 $7 >> $O/pmc_sq_$run.log 2>&1
}
for p in modulate demod_mf demod_mf_ic2 demod_zf_ic2; do pmc_sq $p 4096 40 40 64 9 2; pmc_sq $p 65536 12 3 64 9 2; done
pmc_sq demod_mf_ic2 8192 10 2 128 15 4; pmc_sq demod_mf 8192 10 2 256 31 2; pmc_sq demod_zf 8192 10 2 256 31 2
python3 $R/scratch/pmc_summary.py $O/pmc_sq > $O/pmc_sq_counters_summary.csv 2>&1
# 4. kernel traces of the cfg4 / cfg5 shapes at their per-GPU batch (8192) and at 65536, every path
for shape in "128 15 4" "256 31 2"; do for b in 8192 65536; do
  tag=$(echo $shape | tr ' ' '_')_$b
  for p in modulate demod_mf demod_zf demod_mf_ic2 demod_zf_ic2; do
    rocprofv3 --kernel-trace --output-format csv -d $O/shape_trace/$tag/$p -o t -- python3 $R/scratch/run_kernel.py $p $b 40 2 $shape > /dev/null 2>&1
  done
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
# 5. event-timed per-shape tables and the fused transmitter / frame receiver / estimator scripts
python3 $R/scratch/bench_shape.py 64 9 2 65536 > $O/shape_64_9_2_65536.txt 2>&1
python3 $R/scratch/bench_shape.py 32 5 2 65536 0.5 > $O/shape_32_5_2_65536.txt 2>&1
python3 $R/scratch/bench_shape.py 128 15 4 8192 > $O/shape_128_15_4_8192.txt 2>&1
python3 $R/scratch/bench_shape.py 128 15 4 65536 > $O/shape_128_15_4_65536.txt 2>&1
python3 $R/scratch/bench_shape.py 256 31 2 8192 0.1 > $O/shape_256_31_2_8192.txt 2>&1
python3 $R/scratch/bench_shape.py 128 21 2 4096 0.35 > $O/shape_128_21_2_4096.txt 2>&1
python3 $R/scratch/bench_shape.py 16 7 2 65536 0.3 > $O/shape_16_7_2_65536_jit.txt 2>&1
python3 $R/scratch/bench_shape.py 96 25 2 4096 0.35 > $O/shape_96_25_2_4096.txt 2>&1
python3 $R/scratch/bench_shape.py 16 127 2 4096 0.5 > $O/shape_16_127_2_4096_generic.txt 2>&1
python3 $R/scratch/bench_shape.py 1024 15 2 2048 > $O/shape_1024_15_2_2048_jit.txt 2>&1
python3 $R/scratch/bench_stages.py > $O/bench_stages.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/est -o est -- python3 $R/scratch/bench_est.py > $O/bench_est.txt 2>&1
python3 $R/scratch/bench_tx.py > $O/bench_tx.txt 2>&1
python3 $R/scratch/bench_frames.py > $O/bench_frames.txt 2>&1
# keep the summaries, drop the raw profiler output (gpurun copies back at most 64 MiB)
python3 $R/scratch/trace_by_shape.py $O/bench/bench_kernel_trace.csv | grep -E "^kernel|k_row|k_generic|k_est" > $O/bench_default_kernel_durations_by_launch_shape.csv
python3 $R/scratch/trace_by_shape.py $O/est/est_kernel_trace.csv | grep -E "^kernel|k_row|k_generic|k_est" > $O/bench_est_kernel_durations_by_launch_shape.csv
cp $O/bench/bench_kernel_stats.csv $O/bench_default_kernel_stats.csv; cp $O/bench/bench_domain_stats.csv $O/bench_default_domain_stats.csv
cp $O/est/est_kernel_stats.csv $O/bench_est_kernel_stats.csv
rm -rf $O/pmc_hbm $O/pmc_sq $O/shape_trace $O/bench $O/est $O/*.log
du -sh $O; ls $O | head -80
