#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3w; mkdir -p $O
cd $R
timeout 120 scratch/probe/wave_dft31 > $O/wave_dft31.txt 2>&1; echo "rc=$?" >> $O/wave_dft31.txt; cat $O/wave_dft31.txt
