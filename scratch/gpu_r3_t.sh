#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3t; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/all.txt 2>&1; echo "rc=$?" >> $O/all.txt; tail -6 $O/all.txt
for round in 1 2; do for on in 1 0; do
  for sh in "16 127 2 4096 0.5" "32 100 2 4096 0.3" "37 21 2 4096 0.35" "61 7 2 4096 0.3" "16 63 2 4096 0.3" "37 127 2 2048 0.3"; do echo "== tree mx$on $sh"; GFDM_DFT_MX=$on python3 scratch/bench_shape.py $sh 2>&1 | grep -E "modulate|demod_mf |demod_zf |demod_zf_ic2"; done
done; done | tee $O/ab9.txt
