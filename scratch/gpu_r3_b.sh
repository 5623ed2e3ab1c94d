#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3a; mkdir -p $O
cd $R
timeout 900 python -X faulthandler -m pytest tests -x -q -m gpu -k "matrix_cores or tie_inputs or fused_estimator or sharded or critical_path or prefixer or argument_errors" > $O/sel.txt 2>&1; echo "sel rc=$?" >> $O/sel.txt; tail -15 $O/sel.txt
scratch/ab_k128.sh "tree single waves5 single5 nopfa"
