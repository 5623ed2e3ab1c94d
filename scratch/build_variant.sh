#!/bin/bash
# Build libgfdm_hip.so with extra compile flags into scratch/ab/<tag>/ (A/B timing on one box: GFDM_HIP_LIB=<that file>).
#   scratch/build_variant.sh <tag> [-DFLAG ...]
set -e
tag=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
make -C $R/gr-gfdm_amd -j8 hip OUT=$R/scratch/ab/$tag HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize $*" > $R/scratch/ab/$tag.log 2>&1 || { tail -20 $R/scratch/ab/$tag.log; exit 1; }
ls -la $R/scratch/ab/$tag/libgfdm_hip.so
