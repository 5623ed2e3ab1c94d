#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu -k "generic or every_entry or advanced_receiver_against or golden or larger_than or stages or transmitter or estimator" > $O/sel.txt 2>&1; echo "rc=$?" >> $O/sel.txt; tail -5 $O/sel.txt
for round in 1 2; do for tag in tree prev; do
  if [ "$tag" = "tree" ]; then unset GFDM_HIP_LIB; else export GFDM_HIP_LIB=$R/scratch/ab/$tag/libgfdm_hip.so; fi
  for sh in "37 21 2 4096 0.35" "61 7 2 4096 0.3" "50 9 2 4096 0.3" "66 5 2 4096 0.3" "99 7 2 4096 0.3" "16 127 2 4096 0.5"; do echo "== $tag $sh"; python3 scratch/bench_shape.py $sh 2>&1 | grep -E "modulate|demod_mf |demod_zf_ic2"; done
done; done | tee $O/generic_ab3.txt
