#!/bin/bash
# A/B of variant builds on one shape with scratch/bench_shape.py (event-timed):  ab_shape.sh "<tag> ..." K M L B [alpha]   ("" = the in-tree build)
R=$GRAFT_REPO_ROOT
tags=$1; shift
for round in 1 2; do for tag in $tags; do
  if [ "$tag" = "tree" ]; then unset GFDM_HIP_LIB; else export GFDM_HIP_LIB=$R/scratch/ab/$tag/libgfdm_hip.so; fi
  echo "== $tag (round $round)"; python3 $R/scratch/bench_shape.py "$@" 2>&1 | grep -v amdgpu.ids
done; done
