#!/bin/bash
# Round-5 GPU runner:   gpurun -- bash scratch/gpu_r5.sh <task> [tag]      (tasks of round 4 that are still used were moved here)
#   newtests   the tests this round added (stated 65 536-block batches, duplicate subcarrier-map entry), then nothing else
#   tests      the whole -m gpu suite + smoke
#   probe      scratch/sustained_probe.py on the kernels whose back-to-back and per-launch figures disagreed in round 4
#   bench      bench.py default + cfg3 / cfg4 / cfg5 + the driver's command line
#   alone      rocprofv3 kernel durations, one kernel on the GPU at a time
#   pmc        FETCH_SIZE / WRITE_SIZE per kernel (separate passes), summary with the build id
R=$GRAFT_REPO_ROOT; T=${2:-r5}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
trace() {   # trace <out.csv> <label> <min launches> <run_kernel args...>
  local out=$1 label=$2 minl=$3; shift 3
  rm -rf /tmp/alone/$label
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/alone/$label -o t -- python3 $R/scratch/run_kernel.py "$@" > /dev/null 2>&1
  python3 $R/scratch/trace_by_shape.py /tmp/alone/$label/t_kernel_trace.csv | grep -E "k_row|k_est|k_generic|k_rader" | awk -v l=$label -v r=$minl -F'"' '{split($3,a,","); if (a[5]+0 >= r) print l "," "\"" $2 "\"" $3}' >> $out
}
case $1 in
newtests)
  cd $R
  timeout 1500 python -m pytest tests/test_stated_batch_gpu.py tests/test_parity_gpu.py -x -q -k "stated or duplicate" > $O/pytest_new.txt 2>&1; echo "rc=$?" >> $O/pytest_new.txt; tail -15 $O/pytest_new.txt ;;
tests)
  cd $R
  timeout 2700 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt; tail -14 $O/pytest_gpu.txt
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
probe)     # PROBE_SPECS: "path blocks K M L" entries separated by ';' (default: the kernels of VERDICT r04 weak item 4 and the 8-GPU configs)
  specs=${PROBE_SPECS:-"demod_zf_ic2 65536 64 9 2;demod_mf_ic2 65536 64 9 2;demod_mf 65536 64 9 2;demod_zf_ic2 4096 64 9 2;demod_mf_ic2 8192 128 15 4;demod_mf_ic2 65536 128 15 4;demod_zf 8192 256 31 2;demod_zf 65536 256 31 2"}
  IFS=';' read -ra list <<< "$specs"
  for spec in "${list[@]}"; do
    set -- $spec
    timeout 300 python3 $R/scratch/sustained_probe.py $1 $2 $3 $4 $5 2.0 > $O/sustained_$1_$3_$4_$5_$2.txt 2>&1
    cut -c1-260 $O/sustained_$1_$3_$4_$5_$2.txt
  done ;;
bench)
  cd $R
  python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
  for c in cfg3 cfg4 cfg5; do python3 bench.py --config $c --no-cpu-baseline --no-host-paths > $O/bench_$c.json 2>/dev/null; cut -c1-200 $O/bench_$c.json; done
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2>/dev/null; cut -c1-200 $O/bench_driver_line.json ;;
alone)
  python3 -c "import sys; sys.path.insert(0, '$R/gr-gfdm_amd/python'); import gfdm_amd; print(gfdm_amd.build_id())" > $O/build_id.txt
  echo "label,kernel,workgroups,workgroup_size,queue,launches,mean_us,median_us,min_us,max_us" > $O/kernel_alone.csv
  for B in 4096 65536; do
    reps=400; slots=36; [ $B = 65536 ] && { reps=60; slots=3; }
    for p in modulate demod_mf demod_zf demod_mf_ic2 demod_zf_ic2; do trace $O/kernel_alone.csv 64_9_2_${p}_$B $((reps / 2)) $p $B $reps $slots; done
  done
  for b in 8192 65536; do
    for p in demod_mf demod_mf_ic2 demod_zf_ic2; do trace $O/kernel_alone.csv 128_15_4_${p}_$b 20 $p $b 40 2 128 15 4; done
    for p in demod_mf demod_zf modulate; do trace $O/kernel_alone.csv 256_31_2_${p}_$b 10 $p $b 20 2 256 31 2; done
  done
  for b in 4096 65536; do   # the reference's QA shape (127 timeslots, 16 subcarriers): Rader kernels, and the dense matrix-core form beside them (GFDM_DFT_MX=2)
    reps=100; slots=12; [ $b = 65536 ] && { reps=20; slots=2; }
    for p in modulate demod_mf demod_zf demod_mf_ic2 demod_zf_ic2; do trace $O/kernel_alone.csv 16_127_2_${p}_$b $((reps / 2)) $p $b $reps $slots 16 127 2; done
    for p in modulate demod_mf demod_zf demod_mf_ic2 demod_zf_ic2; do GFDM_DFT_MX=2 trace $O/kernel_alone.csv 16_127_2_dense_${p}_$b $((reps / 2)) $p $b $reps $slots 16 127 2; done
  done
  cat $O/kernel_alone.csv | cut -d, -f1-2,8- ;;
pmc)
  rm -rf $O/pmc; mkdir -p $O/pmc
  id=$(python3 -c "import sys; sys.path.insert(0, '$R/gr-gfdm_amd/python'); import gfdm_amd; print(gfdm_amd.build_id())")
  for spec in "modulate 4096 64 9 2" "demod_mf 4096 64 9 2" "demod_zf 4096 64 9 2" "demod_mf_ic2 4096 64 9 2" "demod_zf_ic2 4096 64 9 2" "demod_zf_ic2 65536 64 9 2" \
              "demod_mf_ic2 8192 128 15 4" "demod_mf_ic2 65536 128 15 4" "demod_zf 8192 256 31 2" "demod_zf 65536 256 31 2" \
              "demod_mf 4096 16 127 2" "modulate 4096 16 127 2" "demod_zf 4096 16 127 2" "demod_zf_ic2 4096 16 127 2"; do
    set -- $spec; run=$1_$3_$4_$5_$2; reps=40; [ $2 -ge 65536 ] && reps=12; [ $2 -ge 65536 ] && [ $3 -ge 256 ] && reps=6
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc/$run/$c -o pmc -- python3 $R/scratch/run_kernel.py $1 $2 $reps 2 $3 $4 $5 > /dev/null 2>&1
    done
  done
  python3 $R/scratch/pmc_summary.py $O/pmc $id > $O/pmc_hbm_traffic_summary.csv; rm -rf $O/pmc
  grep -E "k_row_receive|^run" $O/pmc_hbm_traffic_summary.csv | head -40 ;;
*) echo "unknown task $1"; exit 2 ;;
esac
