#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3v; mkdir -p $O
cd $R
L=$R/scratch/ab/gstamps/libgfdm_hip.so
for args in "demod_mf 8192 256 31 2" "demod_mf 512 256 31 2" "demod_zf 8192 256 31 2" "demod_mf 8192 128 15 4"; do
  python3 scratch/stamps.py $L $args 2>&1 | grep -v amdgpu
done | tee $O/stamps_256.txt
