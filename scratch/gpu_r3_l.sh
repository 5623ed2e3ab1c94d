#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3k; mkdir -p $O
cd $R
for round in 1 2; do
for tag in tree prev; do
  if [ "$tag" = "tree" ]; then unset GFDM_HIP_LIB; else export GFDM_HIP_LIB=$R/scratch/ab/$tag/libgfdm_hip.so; fi
  echo "== $tag"; python scratch/host_latency.py 2>&1 | grep -v amdgpu | head -4
done; done | tee $O/host_latency_ab.txt
