#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3a; mkdir -p $O
cd $R
rm -f $O/errlog.txt
GFDM_ERRLOG=$O/errlog.txt timeout 2400 python -X faulthandler -m pytest tests -x -q -m gpu > $O/all.txt 2>&1; echo "all rc=$?" >> $O/all.txt; tail -6 $O/all.txt
python scratch/errlog_table.py $O/errlog.txt > $O/errtable.md; cat $O/errtable.md
