#!/bin/bash
# Build libgfdm_hip.so of an earlier commit into scratch/ab/<tag>/ (A/B against the working tree on one box):  scratch/build_rev.sh <tag> <git rev>
set -e
tag=$1; rev=$2
R=$(cd $(dirname $0)/.. && pwd)
rm -rf /tmp/wt_$tag; git -C $R worktree add -f /tmp/wt_$tag $rev > /dev/null 2>&1
mkdir -p $R/scratch/ab
make -C /tmp/wt_$tag/gr-gfdm_amd -j8 hip OUT=$R/scratch/ab/$tag > $R/scratch/ab/$tag.log 2>&1 || { tail -20 $R/scratch/ab/$tag.log; exit 1; }
git -C $R worktree remove --force /tmp/wt_$tag
ls -la $R/scratch/ab/$tag/libgfdm_hip.so
