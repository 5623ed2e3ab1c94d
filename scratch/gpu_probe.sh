#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3p
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/host_latency scratch/probe/host_latency.hip 2>/dev/null && /tmp/host_latency | tee gpurun_out/r3p/host_latency.txt
