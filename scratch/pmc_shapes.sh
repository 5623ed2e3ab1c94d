#!/bin/bash
# SQ counters + kernel trace of the cfg4 / cfg5 shapes at their per-GPU batch (8192 blocks):  gpurun -- scratch/pmc_shapes.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_shapes
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # path K M L
  tag=$1_$2_$3_$4
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/${tag}_8192_a -o pmc -- python3 $R/scratch/run_kernel.py $1 8192 12 2 $2 $3 $4 > $O/${tag}_a.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $O/${tag}_8192_b -o pmc -- python3 $R/scratch/run_kernel.py $1 8192 12 2 $2 $3 $4 > $O/${tag}_b.log 2>&1
  rocprofv3 --kernel-trace --output-format csv -d $O/${tag}_8192_t -o trace -- python3 $R/scratch/run_kernel.py $1 8192 40 2 $2 $3 $4 > $O/${tag}_t.log 2>&1
}
for p in modulate demod_mf demod_zf demod_mf_ic2 demod_zf_ic2; do run $p 128 15 4; run $p 256 31 2; done
python3 $R/scratch/pmc_summary.py $O > $O/summary.csv 2>&1
for f in $O/*_t/trace_kernel_trace.csv; do echo $f; python3 $R/scratch/trace_by_shape.py $f | grep k_row; done > $O/trace_summary.txt
cat $O/trace_summary.txt
