#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3k; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu -k "golden or boundary or wrappers or host or threads or ragged" > $O/sel.txt 2>&1; echo "rc=$?" >> $O/sel.txt; tail -5 $O/sel.txt
python scratch/host_latency.py 2>&1 | grep -v amdgpu | tee $O/host_latency.txt
