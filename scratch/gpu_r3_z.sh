#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3z; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/all.txt 2>&1; echo "rc=$?" >> $O/all.txt; tail -4 $O/all.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-400
python3 bench.py --gpus 2 --dist-backend gloo --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-400
