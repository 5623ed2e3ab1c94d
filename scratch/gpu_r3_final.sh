#!/bin/bash
# final-tree check of the round: full GPU suite with the error log, smoke, single-kernel durations + bench headline on this box (tag finaltree), bench JSON lines, generic stamps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3final; mkdir -p $O
cd $R
rm -f $O/errlog.txt
GFDM_ERRLOG=$O/errlog.txt timeout 2400 python -m pytest tests -x -q -m gpu > $O/all.txt 2>&1; echo "rc=$?" >> $O/all.txt; tail -4 $O/all.txt
python3 scratch/errlog_table.py $O/errlog.txt > $O/parity_error_table.md 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
scratch/profile_alone3.sh finaltree > $O/alone.txt 2>&1; tail -22 $O/alone.txt
cd $R
python3 bench.py > $O/bench_default_final_tree.json 2> $O/bench_default.err; cat $O/bench_default_final_tree.json | cut -c1-600
for c in cfg3 cfg4 cfg5; do python3 bench.py --config $c --cpu-seconds 3 > $O/bench_${c}_final_tree.json 2>/dev/null; done
L=$R/scratch/ab/gstamps/libgfdm_hip.so
for args in "256 16 127 2 1 modulate" "4096 16 127 2 1 modulate" "256 16 127 2 1 demod_mf" "4096 16 127 2 1 demod_mf" "4096 16 127 2 0 modulate"; do
  python3 scratch/stamps_generic.py $L $args 2>&1 | grep -v amdgpu
done > $O/stamps_generic_16_127.txt
python3 scratch/bench_shape.py 16 127 2 4096 0.5 2>&1 | grep -v amdgpu > $O/shape_16_127_2_4096_generic.txt
cp $R/gpurun_out/round3_finaltree/*.csv $R/gpurun_out/round3_finaltree/*.json $O/ 2>/dev/null
