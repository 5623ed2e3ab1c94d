#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE in separate passes) of one path / shape:  gpurun -- bash scratch/pmc_hbm_one.sh <path> <batch> <K> <M> <L>
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_one
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run=$1_$3_$4_$5_$2
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/$run/$c -o pmc -- python3 $R/scratch/run_kernel.py $1 $2 12 2 $3 $4 $5 > $O/$c.log 2>&1
done
python3 $R/scratch/pmc_summary.py $O > $O/summary.csv 2>&1
grep -E "k_row|^run" $O/summary.csv
rm -rf $O/$run
