#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3i; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu -k "critical_path or estimator or multi_rank" > $O/sel.txt 2>&1; echo "rc=$?" >> $O/sel.txt; tail -8 $O/sel.txt
