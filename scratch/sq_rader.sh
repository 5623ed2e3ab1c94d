R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq_rader; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for p in demod_mf modulate; do
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O/$p/a -o pmc -- python3 $R/scratch/run_kernel.py $p 4096 40 2 16 127 2 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA --output-format csv -d $O/$p/b -o pmc -- python3 $R/scratch/run_kernel.py $p 4096 40 2 16 127 2 > /dev/null 2>&1
done
python3 $R/scratch/pmc_summary.py $O x | grep -i rader
