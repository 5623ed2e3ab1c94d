#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3a
python scratch/dbg_jit.py 16 256 > gpurun_out/r3a/dbg.txt 2>&1; echo "rc=$?" >> gpurun_out/r3a/dbg.txt
python scratch/dbg_jit.py 3 32 >> gpurun_out/r3a/dbg.txt 2>&1; echo "rc=$?" >> gpurun_out/r3a/dbg.txt
cat gpurun_out/r3a/dbg.txt
