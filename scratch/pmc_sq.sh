#!/bin/bash
# SQ / LDS counters of the final row-lane kernels at the benchmark batch (4096) and at 65 536 blocks:  gpurun -- scratch/pmc_sq.sh
# (counters in their own runs: --pmc only, program directly after "--"; SQ has 8 slots per pass on gfx950)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_sq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_available.txt 2>&1
for path in modulate demod_mf demod_zf_ic2 demod_mf_ic2; do
  for B in 4096 65536; do
    reps=40; [ $B = 65536 ] && reps=12
    rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/${path}_${B}_a -o pmc -- python3 $R/scratch/run_kernel.py $path $B $reps > $O/${path}_${B}_a.log 2>&1
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $O/${path}_${B}_b -o pmc -- python3 $R/scratch/run_kernel.py $path $B $reps > $O/${path}_${B}_b.log 2>&1
  done
done
python3 $R/scratch/pmc_summary.py $O > $O/summary.csv 2>&1
tail -3 $O/*_a.log | head -60
