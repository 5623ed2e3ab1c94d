#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3o; mkdir -p $O
cd $R
scratch/probe/mfma_f32_layout > $O/probe.txt 2>&1; cat $O/probe.txt
timeout 1500 python -m pytest tests -x -q -m gpu -k "matrix_core_timeslot or larger_than or generic" > $O/sel.txt 2>&1; echo "rc=$?" >> $O/sel.txt; tail -15 $O/sel.txt
for on in 1 0; do
  for sh in "16 127 2 4096 0.5" "16 63 2 4096 0.3" "37 127 2 2048 0.3" "8 40 2 4096 0.3"; do echo "== mx$on $sh"; GFDM_DFT_MX=$on python3 scratch/bench_shape.py $sh 2>&1 | grep -E "modulate|demod_mf |demod_zf |demod_zf_ic2|kernel"; done
done | tee $O/mx_ab.txt
