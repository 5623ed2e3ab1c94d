#!/bin/bash
# SQ counters of the two-pass row-lane kernels for subcarrier counts that are not a power of two:  gpurun -- bash scratch/pmc_mixed.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_mixed
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # path K M L B
  tag=$1_$2_$3_$4
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/${tag}_$5_a -o pmc -- python3 $R/scratch/run_kernel.py $1 $5 12 2 $2 $3 $4 > $O/${tag}_a.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $O/${tag}_$5_b -o pmc -- python3 $R/scratch/run_kernel.py $1 $5 12 2 $2 $3 $4 > $O/${tag}_b.log 2>&1
  rocprofv3 --kernel-trace --output-format csv -d $O/${tag}_$5_t -o trace -- python3 $R/scratch/run_kernel.py $1 $5 40 2 $2 $3 $4 > $O/${tag}_t.log 2>&1
  rm -rf $O/${tag}_$5_*/*/*.db 2>/dev/null
}
run demod_mf 96 25 2 4096
run demod_mf 48 9 2 16384
run demod_mf 12 5 2 131072
run demod_zf_ic2 96 25 2 4096
python3 $R/scratch/pmc_summary.py $O > $O/summary.csv 2>&1
for f in $O/*_t/trace_kernel_trace.csv; do echo $f; python3 $R/scratch/trace_by_shape.py $f | grep k_row; done > $O/trace_summary.txt
cat $O/trace_summary.txt; grep -E "SQ_WAVES|SQ_WAVE_CYCLES|SQ_LDS_BANK_CONFLICT|SQ_INSTS_LDS|SQ_INSTS_VALU|SQ_WAIT_INST_LDS|SQ_ACTIVE_INST_ANY|SQ_WAIT_ANY,|SQ_WAIT_INST_ANY" $O/summary.csv | grep k_row_receive
du -sh $O
