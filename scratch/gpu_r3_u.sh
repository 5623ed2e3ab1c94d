#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3u; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu -k "fused_estimator or phase_compensation or matrix_core" > $O/sel.txt 2>&1; echo "rc=$?" >> $O/sel.txt; tail -15 $O/sel.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
