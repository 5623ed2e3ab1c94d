#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu -k "matrix_core_timeslot or larger_than or generic or phase_comp or golden" > $O/sel.txt 2>&1; echo "rc=$?" >> $O/sel.txt; tail -12 $O/sel.txt
for round in 1 2; do
  for on in 2 1 0; do
  for sh in "16 127 2 4096 0.5" "16 63 2 4096 0.3" "61 33 2 4096 0.3" "37 127 2 2048 0.3" "32 100 2 4096 0.3"; do echo "== tree mx$on $sh"; GFDM_DFT_MX=$on python3 scratch/bench_shape.py $sh 2>&1 | grep -E "modulate|demod_mf |demod_zf_ic2"; done
  done
done | tee $O/ab5.txt
