#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; mkdir -p $O
cd $R
for p in demod_mf demod_mf_ic2; do python scratch/stamps.py $R/scratch/ab/stamps/libgfdm_hip.so $p 8192 128 15 4 2>&1 | grep -v amdgpu.ids; done | tee $O/stamps_128.txt
for p in demod_mf demod_zf_ic2; do python scratch/stamps.py $R/scratch/ab/stamps/libgfdm_hip.so $p 4096 64 9 2 2>&1 | grep -v amdgpu.ids; done | tee $O/stamps_64.txt
cd /tmp && export TMPDIR=/tmp
pmc_sq() {  # path batch reps slots K M L
  run=$1_$5_$6_$7_$2
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq/$run/a -o pmc -- python3 $R/scratch/run_kernel.py $1 $2 $3 $4 $5 $6 $7 > $O/pmc_sq_$run.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_sq/$run/b -o pmc -- python3 $R/scratch/run_kernel.py $1 $2 $3 $4 $5 $6 $7 >> $O/pmc_sq_$run.log 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_WAIT_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_F16 SQ_ACTIVE_INST_MISC --output-format csv -d $O/pmc_sq/$run/c -o pmc -- python3 $R/scratch/run_kernel.py $1 $2 $3 $4 $5 $6 $7 >> $O/pmc_sq_$run.log 2>&1
}
pmc_sq demod_mf 8192 10 2 128 15 4; pmc_sq demod_mf_ic2 8192 10 2 128 15 4
GFDM_MX=0 pmc_sq demod_mf_ic2 8192 10 2 128 15 4 && mv $O/pmc_sq/demod_mf_ic2_128_15_4_8192 $O/pmc_sq/demod_mf_ic2valu_128_15_4_8192
pmc_sq demod_mf_ic2 8192 10 2 128 15 4
python3 $R/scratch/pmc_summary.py $O/pmc_sq > $O/pmc_sq_summary.csv 2>&1
rm -rf $O/pmc_sq; cat $O/pmc_sq_summary.csv | cut -d, -f1-5
