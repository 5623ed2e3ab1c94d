#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3s; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu -k "matrix_core_timeslot or larger_than or generic or fused_estimator or phase_compensation" > $O/sel.txt 2>&1; echo "rc=$?" >> $O/sel.txt; tail -5 $O/sel.txt
L=$R/scratch/ab/gstamps/libgfdm_hip.so
for args in "256 16 127 2 1 modulate" "256 16 127 2 1 demod_mf" "4096 16 127 2 1 modulate"; do
  python3 scratch/stamps_generic.py $L $args 2>&1 | grep -v amdgpu
done | tee $O/stamps_generic5.txt
for round in 1 2; do for on in 1; do
  for sh in "16 127 2 4096 0.5" "32 100 2 4096 0.3"; do echo "== tree mx$on $sh"; GFDM_DFT_MX=$on python3 scratch/bench_shape.py $sh 2>&1 | grep -E "modulate|demod_mf |demod_zf |demod_zf_ic2"; done
done; done | tee $O/ab10.txt
