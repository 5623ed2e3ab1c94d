#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3fuzz; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 1200 python3 scratch/fuzz_shapes.py 77 420 > $O/fuzz_all3.txt 2>&1; echo "rc=$? after $(( $(date +%s) - t0 )) s" >> $O/fuzz_all3.txt; tail -4 $O/fuzz_all3.txt
