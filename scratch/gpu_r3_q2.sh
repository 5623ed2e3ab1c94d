#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q2; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu -k "process_exit or jit or background or damaged or run_time or critical" > $O/sel.txt 2>&1; echo "rc=$?" >> $O/sel.txt; tail -25 $O/sel.txt
