#!/bin/bash
# Round-4 GPU runner (replaces the one-off scratch/gpu_r3_*.sh launchers):   gpurun -- bash scratch/gpu_r4.sh <task> [tag]
# (every profiler run sits under its own `timeout`: one collection of the round hung at a profiled process's exit and ran into gpurun's limit)
#   tests      the whole -m gpu suite + smoke
#   bench      bench.py default + cfg3 / cfg4 / cfg5 headline lines
#   events     HIP-event timings of single kernels beside rocprofv3 --kernel-trace on the same box
#   alone      rocprofv3 kernel durations, one kernel on the GPU at a time (K=64 M=9 paths, K=128 M=15 L=4 and K=256 M=31 paths)
#   pmc        FETCH_SIZE / WRITE_SIZE per kernel (separate passes), summary with the build id
R=$GRAFT_REPO_ROOT; T=${2:-r4}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
trace() {   # trace <out.csv> <label> <min launches> <run_kernel args...>
  local out=$1 label=$2 minl=$3; shift 3
  rm -rf /tmp/alone/$label
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/alone/$label -o t -- python3 $R/scratch/run_kernel.py "$@" > /dev/null 2>&1
  python3 $R/scratch/trace_by_shape.py /tmp/alone/$label/t_kernel_trace.csv | grep -E "k_row|k_est|k_generic" | awk -v l=$label -v r=$minl -F'"' '{split($3,a,","); if (a[5]+0 >= r) print l "," "\"" $2 "\"" $3}' >> $out
}
case $1 in
tests)
  cd $R
  timeout 2700 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt; tail -5 $O/pytest_gpu.txt
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
bench)
  cd $R
  python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
  for c in cfg3 cfg4 cfg5; do python3 bench.py --config $c --no-cpu-baseline --no-host-paths > $O/bench_$c.json 2>/dev/null; cut -c1-200 $O/bench_$c.json; done
  # the driver's own command line (a 20-step burst = 0.3 ms timed region)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2>/dev/null; cut -c1-200 $O/bench_driver_line.json ;;
events)
  python3 $R/scratch/event_timing_probe.py > $O/event_timing_probe.txt 2>&1; cat $O/event_timing_probe.txt
  echo "label,kernel,workgroups,workgroup_size,queue,launches,mean_us,median_us,min_us,max_us" > $O/event_probe_rocprof.csv
  for p in modulate demod_mf demod_zf_ic2; do trace $O/event_probe_rocprof.csv ${p}_4096 200 $p 4096 400 36; done
  cat $O/event_probe_rocprof.csv ;;
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
  cat $O/kernel_alone.csv | cut -d, -f1-2,8- ;;
pmc)
  rm -rf $O/pmc; mkdir -p $O/pmc
  id=$(python3 -c "import sys; sys.path.insert(0, '$R/gr-gfdm_amd/python'); import gfdm_amd; print(gfdm_amd.build_id())")
  for spec in "modulate 4096 64 9 2" "demod_mf 4096 64 9 2" "demod_zf 4096 64 9 2" "demod_mf_ic2 4096 64 9 2" "demod_zf_ic2 4096 64 9 2" "demod_zf_ic2 65536 64 9 2" \
              "demod_mf_ic2 8192 128 15 4" "demod_mf_ic2 65536 128 15 4" "demod_zf 8192 256 31 2" "demod_zf 65536 256 31 2"; do
    set -- $spec; run=$1_$3_$4_$5_$2; reps=40; [ $2 -ge 65536 ] && reps=12; [ $2 -ge 65536 ] && [ $3 -ge 256 ] && reps=6
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc/$run/$c -o pmc -- python3 $R/scratch/run_kernel.py $1 $2 $reps 2 $3 $4 $5 > /dev/null 2>&1
    done
  done
  python3 $R/scratch/pmc_summary.py $O/pmc $id > $O/pmc_hbm_traffic_summary.csv; rm -rf $O/pmc
  grep -E "k_row_receive|^run" $O/pmc_hbm_traffic_summary.csv | head -40 ;;
sq)     # SQ / LDS counters per wave (two passes of eight counters), with the build id
  rm -rf $O/sq; mkdir -p $O/sq
  id=$(python3 -c "import sys; sys.path.insert(0, '$R/gr-gfdm_amd/python'); import gfdm_amd; print(gfdm_amd.build_id())")
  for spec in "modulate 4096 64 9 2 1" "demod_mf 4096 64 9 2 1" "demod_mf_ic2 4096 64 9 2 1" "demod_zf_ic2 4096 64 9 2 1" "demod_zf_ic2 65536 64 9 2 1" \
              "demod_mf 8192 128 15 4 1" "demod_mf_ic2 8192 128 15 4 1" "demod_mf_ic2 8192 128 15 4 0" "demod_zf 8192 256 31 2 1"; do
    set -- $spec; run=$1_$3_$4_$5_$2; [ $6 = 0 ] && run=${run}_valu; reps=40; [ $2 -ge 65536 ] && reps=12
    GFDM_MX=$6 timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq/$run/a -o pmc -- python3 $R/scratch/run_kernel.py $1 $2 $reps 2 $3 $4 $5 > /dev/null 2>&1
    GFDM_MX=$6 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA --output-format csv -d $O/sq/$run/b -o pmc -- python3 $R/scratch/run_kernel.py $1 $2 $reps 2 $3 $4 $5 > /dev/null 2>&1
  done
  python3 $R/scratch/pmc_summary.py $O/sq $id > $O/pmc_sq_counters_summary.csv; rm -rf $O/sq
  grep -c . $O/pmc_sq_counters_summary.csv ;;
ic)     # the interference-cancellation kernels: parity first, then durations with the rounds on the matrix cores / vector ALU
  cd $R
  timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -k "ic or matrix or zero or golden or full_size" > $O/pytest_ic.txt 2>&1; echo "rc=$?" >> $O/pytest_ic.txt; tail -4 $O/pytest_ic.txt
  cd /tmp
  echo "label,kernel,workgroups,workgroup_size,queue,launches,mean_us,median_us,min_us,max_us" > $O/ic_kernels.csv
  for mx in 1 0 2; do
    for b in 8192 65536; do GFDM_MX=$mx trace $O/ic_kernels.csv 128_15_4_mx${mx}_mf_ic2_$b 20 demod_mf_ic2 $b 40 2 128 15 4; GFDM_MX=$mx trace $O/ic_kernels.csv 128_15_4_mx${mx}_zf_ic2_$b 20 demod_zf_ic2 $b 40 2 128 15 4; done
    for b in 4096 65536; do reps=400; slots=36; [ $b = 65536 ] && { reps=60; slots=3; }
      GFDM_MX=$mx trace $O/ic_kernels.csv 64_9_2_mx${mx}_mf_ic2_$b $((reps / 2)) demod_mf_ic2 $b $reps $slots; GFDM_MX=$mx trace $O/ic_kernels.csv 64_9_2_mx${mx}_zf_ic2_$b $((reps / 2)) demod_zf_ic2 $b $reps $slots; done
  done
  trace $O/ic_kernels.csv 128_15_4_mf_8192 20 demod_mf 8192 40 2 128 15 4; trace $O/ic_kernels.csv 128_15_4_mf_65536 20 demod_mf 65536 40 2 128 15 4
  cut -d, -f1-2,7-9 $O/ic_kernels.csv ;;
icab)   # same-box A/B of the cancellation kernels: working tree, the register-transpose variant (scratch/ab/regx), round 3 (scratch/ab/r3 + its package)
  echo "label,kernel,workgroups,workgroup_size,queue,launches,mean_us,median_us,min_us,max_us" > $O/ic_ab.csv
  for rep in 1 2; do
  for v in tree regx r3; do
    unset GFDM_HIP_LIB GFDM_PKG
    [ $v = regx ] && export GFDM_HIP_LIB=$R/scratch/ab/regx/libgfdm_hip.so
    [ $v = r3 ] && export GFDM_HIP_LIB=$R/scratch/ab/r3/libgfdm_hip.so GFDM_PKG=$R/scratch/old_pkg
    for b in 8192 65536; do trace $O/ic_ab.csv ${v}_128_15_4_mf_ic2_${b}_$rep 20 demod_mf_ic2 $b 40 2 128 15 4; trace $O/ic_ab.csv ${v}_128_15_4_zf_ic2_${b}_$rep 20 demod_zf_ic2 $b 40 2 128 15 4; done
    GFDM_MX=2 trace $O/ic_ab.csv ${v}_64_9_2_mx2_mf_ic2_4096_$rep 200 demod_mf_ic2 4096 400 36
    GFDM_MX=2 trace $O/ic_ab.csv ${v}_64_9_2_mx2_zf_ic2_4096_$rep 200 demod_zf_ic2 4096 400 36
  done; done
  unset GFDM_HIP_LIB GFDM_PKG
  trace $O/ic_ab.csv tree_64_9_2_dpp_mf_ic2_4096 200 demod_mf_ic2 4096 400 36
  cut -d, -f1,7-9 $O/ic_ab.csv
  cd $R; timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "ic or matrix or zero or golden" > $O/pytest_ic.txt 2>&1; tail -3 $O/pytest_ic.txt ;;
icw)    # same-box A/B of the matrix-core IC kernels: the tree against a variant library scratch/ab/<variant>/ (third argument; parity tests on the variant first)
  V=${3:-w2}
  cd $R; GFDM_HIP_LIB=$R/scratch/ab/$V/libgfdm_hip.so timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "ic or matrix or zero or golden" > $O/pytest_ic_$V.txt 2>&1; tail -3 $O/pytest_ic_$V.txt; cd /tmp
  echo "label,kernel,workgroups,workgroup_size,queue,launches,mean_us,median_us,min_us,max_us" > $O/ic_ab_$V.csv
  for rep in 1 2 3; do
  for v in tree $V; do
    unset GFDM_HIP_LIB
    [ $v = $V ] && export GFDM_HIP_LIB=$R/scratch/ab/$V/libgfdm_hip.so
    for b in 8192 65536; do trace $O/ic_ab_$V.csv ${v}_128_15_4_mf_ic2_${b}_$rep 20 demod_mf_ic2 $b 40 2 128 15 4; trace $O/ic_ab_$V.csv ${v}_128_15_4_zf_ic2_${b}_$rep 20 demod_zf_ic2 $b 40 2 128 15 4; done
    GFDM_MX=2 trace $O/ic_ab_$V.csv ${v}_64_9_2_mx2_mf_ic2_4096_$rep 200 demod_mf_ic2 4096 400 36
  done; done
  unset GFDM_HIP_LIB
  cut -d, -f1,7-9 $O/ic_ab_$V.csv ;;
modab)  # same-box A/B of the K=128 M=15 L=4 modulator: the tree against scratch/ab/<variant>/ (third argument)
  V=${3:-m5}
  cd $R; GFDM_HIP_LIB=$R/scratch/ab/$V/libgfdm_hip.so timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "golden_modulator or full_size" > $O/pytest_mod_$V.txt 2>&1; tail -3 $O/pytest_mod_$V.txt; cd /tmp
  echo "label,kernel,workgroups,workgroup_size,queue,launches,mean_us,median_us,min_us,max_us" > $O/mod_ab_$V.csv
  for rep in 1 2 3; do
  for v in tree $V; do
    unset GFDM_HIP_LIB
    [ $v = $V ] && export GFDM_HIP_LIB=$R/scratch/ab/$V/libgfdm_hip.so
    for b in 8192 65536; do trace $O/mod_ab_$V.csv ${v}_128_15_4_modulate_${b}_$rep 20 modulate $b 40 2 128 15 4; done
  done; done
  unset GFDM_HIP_LIB
  cut -d, -f1,7-9 $O/mod_ab_$V.csv ;;
*) echo "unknown task $1"; exit 2 ;;
esac
