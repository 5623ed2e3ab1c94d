#!/bin/bash
# Round-6 GPU runner:   gpurun -- bash scratch/gpu_r6.sh <task> [tag]
#   icw        VERDICT r05 item 4, the ONE experiment: K=64 M=9 MF / ZF + 2 IC kernels (DPP rounds) built with a register bound (7 / 8 waves per SIMD) against the
#              tree's unbounded build (77-79 registers = 6 waves), rocprofv3 kernel durations, three alternating collections each
#   pkrate     issue costs of the vector instruction classes (round-2 probe)
#   sq         SQ counters (three passes) of the K=64 M=9 MF + 2 IC kernel and its MF sibling at 65 536 / 4096 blocks, and of the Rader kernels of M=127 K=16
#   tests / bench / alone / pmc: as scratch/gpu_r5.sh (delegated)
R=$GRAFT_REPO_ROOT; T=${2:-r6}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
trace() {   # trace <out.csv> <label> <min launches> <run_kernel args...>
  local out=$1 label=$2 minl=$3; shift 3
  rm -rf /tmp/alone/$label
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/alone/$label -o t -- python3 $R/scratch/run_kernel.py "$@" > /dev/null 2>&1
  python3 $R/scratch/trace_by_shape.py /tmp/alone/$label/t_kernel_trace.csv | grep -E "k_row|k_est|k_generic|k_rader" | awk -v l=$label -v r=$minl -F'"' '{split($3,a,","); if (a[5]+0 >= r) print l "," "\"" $2 "\"" $3}' >> $out
}
sq_passes() {   # sq_passes <dir> <run_kernel args...>: the counters in passes of eight
  local d=$1; shift
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $d/a -o pmc -- python3 $R/scratch/run_kernel.py "$@" > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA --output-format csv -d $d/b -o pmc -- python3 $R/scratch/run_kernel.py "$@" > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_LEVEL_WAVES SQ_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $d/c -o pmc -- python3 $R/scratch/run_kernel.py "$@" > $d.c.log 2>&1 || \
    timeout 300 rocprofv3 --pmc SQ_LEVEL_WAVES GRBM_GUI_ACTIVE --output-format csv -d $d/c -o pmc -- python3 $R/scratch/run_kernel.py "$@" > $d.c.log 2>&1
}
case $1 in
icw)
  echo "label,kernel,workgroups,workgroup_size,queue,launches,mean_us,median_us,min_us,max_us" > $O/ic_valu_waves_ab.csv
  for rep in 1 2 3; do
    for v in tree w7 w8; do
      lib=$R/gr-gfdm_amd/lib/libgfdm_hip.so; [ $v != tree ] && lib=$R/scratch/ab/$v/libgfdm_hip.so
      export GFDM_HIP_LIB=$lib
      for p in demod_mf_ic2 demod_zf_ic2; do
        trace $O/ic_valu_waves_ab.csv ${v}_${p}_4096_r$rep 200 $p 4096 400 36
        trace $O/ic_valu_waves_ab.csv ${v}_${p}_65536_r$rep 30 $p 65536 60 3
      done
    done
  done
  unset GFDM_HIP_LIB
  cut -d, -f1,7- $O/ic_valu_waves_ab.csv ;;
dppfold)   # the DPP rotates folded into their one use (v_add_f32_dpp): scratch/ab/dppfold against the library of the tree, parity first, then three alternating collections
  export GFDM_HIP_LIB=$R/scratch/ab/dppfold/libgfdm_hip.so
  (cd $R && timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_stated_batch_gpu.py -x -q -k "advanced or ic or stated or golden" 2>&1 | tail -3)
  unset GFDM_HIP_LIB
  echo "label,kernel,workgroups,workgroup_size,queue,launches,mean_us,median_us,min_us,max_us" > $O/ic_dpp_fold_ab.csv
  for rep in 1 2 3; do
    for v in tree dppfold; do
      lib=$R/gr-gfdm_amd/lib/libgfdm_hip.so; [ $v != tree ] && lib=$R/scratch/ab/$v/libgfdm_hip.so
      export GFDM_HIP_LIB=$lib
      for p in demod_mf_ic2 demod_zf_ic2; do
        trace $O/ic_dpp_fold_ab.csv ${v}_${p}_4096_r$rep 200 $p 4096 400 36
        trace $O/ic_dpp_fold_ab.csv ${v}_${p}_65536_r$rep 30 $p 65536 60 3
      done
    done
  done
  unset GFDM_HIP_LIB
  cut -d, -f1,7- $O/ic_dpp_fold_ab.csv | sed 's/"//g' ;;
icfast)    # clean residency A/B (no spills): the cancellation kernels compiled WITHOUT their cold paths (phase compensation, nearest-point decisions) = 62 / 70 registers = eight /
           # seven waves per SIMD instead of six (scratch/ab/icfast, a timing-only hack of the header), against the library of the tree; also cfg4's shape
  echo "label,kernel,workgroups,workgroup_size,queue,launches,mean_us,median_us,min_us,max_us" > $O/ic_hot_path_only_ab.csv
  for rep in 1 2 3; do
    for v in tree icfast; do
      lib=$R/gr-gfdm_amd/lib/libgfdm_hip.so; [ $v != tree ] && lib=$R/scratch/ab/$v/libgfdm_hip.so
      export GFDM_HIP_LIB=$lib
      for p in demod_mf_ic2 demod_zf_ic2; do
        trace $O/ic_hot_path_only_ab.csv ${v}_${p}_4096_r$rep 200 $p 4096 400 36
        trace $O/ic_hot_path_only_ab.csv ${v}_${p}_65536_r$rep 30 $p 65536 60 3
      done
      trace $O/ic_hot_path_only_ab.csv ${v}_128_15_4_demod_mf_ic2_8192_r$rep 20 demod_mf_ic2 8192 40 2 128 15 4
      trace $O/ic_hot_path_only_ab.csv ${v}_128_15_4_demod_mf_ic2_65536_r$rep 20 demod_mf_ic2 65536 40 2 128 15 4
    done
  done
  unset GFDM_HIP_LIB
  cut -d, -f1,7- $O/ic_hot_path_only_ab.csv | sed 's/"//g' ;;
pkrate)    # issue costs of plain / packed f32, DPP moves and lane swaps at 4 and 16 waves per CU (scratch/probe/pk_rate.hip, the round-2 probe): what the vector-work reading of DESIGN.md section 7 rests on
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/scratch/probe/pk_rate.hip -o /tmp/pk_rate 2>/dev/null && timeout 120 /tmp/pk_rate > $O/valu_issue_costs.txt 2>&1; cat $O/valu_issue_costs.txt ;;
sq)
  rm -rf $O/sq; mkdir -p $O/sq
  id=$(python3 -c "import sys; sys.path.insert(0, '$R/gr-gfdm_amd/python'); import gfdm_amd; print(gfdm_amd.build_id())")
  for spec in "demod_mf_ic2 65536 64 9 2" "demod_mf 65536 64 9 2" "demod_zf_ic2 65536 64 9 2" "demod_mf_ic2 4096 64 9 2" "demod_mf 4096 64 9 2" \
              "demod_mf 4096 16 127 2" "modulate 4096 16 127 2" "demod_zf 4096 16 127 2" "demod_mf_ic2 4096 16 127 2" "demod_zf_ic2 4096 16 127 2"; do
    set -- $spec; run=$1_$3_$4_$5_$2; reps=40; [ $2 -ge 65536 ] && reps=12
    sq_passes $O/sq/$run $1 $2 $reps 2 $3 $4 $5
  done
  python3 $R/scratch/pmc_summary.py $O/sq $id | sed 's/mean_KiB,min_KiB,max_KiB/mean,min,max/' > $O/pmc_sq_counters_summary.csv
  tail -3 $O/sq/*.c.log | tail -20; rm -rf $O/sq
  grep -E "k_row_receive|k_rader" $O/pmc_sq_counters_summary.csv | cut -d, -f1-5 | head -150 ;;
*)
  exec bash $R/scratch/gpu_r5.sh "$@" ;;
esac
