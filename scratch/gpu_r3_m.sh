#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu -k "generic or every_entry or advanced_receiver_against or golden or larger_than or stages or transmitter or estimator" > $O/sel.txt 2>&1; echo "rc=$?" >> $O/sel.txt; tail -5 $O/sel.txt
python3 scratch/bench_shape.py 16 127 2 4096 0.5 2>&1 | grep -v amdgpu | tee $O/shape_16_127_2_4096_generic.txt
python3 scratch/bench_shape.py 37 21 2 4096 0.35 2>&1 | grep -v amdgpu | tee $O/shape_37_21_2_4096_generic.txt
python3 scratch/bench_shape.py 8 63 2 4096 0.35 2>&1 | grep -v amdgpu | tee $O/shape_8_63_2_4096_generic.txt
