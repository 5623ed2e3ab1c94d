#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3fuzz; mkdir -p $O
cd $R
FUZZ_GENERIC=1 timeout 900 python3 scratch/fuzz_shapes.py 31 420 > $O/fuzz_generic.txt 2>&1; tail -15 $O/fuzz_generic.txt
timeout 600 python3 scratch/fuzz_shapes.py 32 240 > $O/fuzz_all.txt 2>&1; tail -5 $O/fuzz_all.txt
