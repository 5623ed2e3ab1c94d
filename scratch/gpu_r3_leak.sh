#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3leak; mkdir -p $O
cd $R
timeout 900 python3 scratch/leak_check.py 400 > $O/leak.txt 2>&1; echo "rc=$?" >> $O/leak.txt; grep -v amdgpu $O/leak.txt | tail -5
