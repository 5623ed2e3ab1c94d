#!/bin/bash
# Everything profiles/<round>/ is assembled from, in one gpurun call:  gpurun -- scratch/profile_round.sh   (then scratch/collect_profiles.sh)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default_plain.json 2> $O/bench_default_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 $R/bench.py > $O/bench_default_under_rocprofv3.json 2> $O/bench_under.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/est -o est -- python3 $R/scratch/bench_est.py > $O/bench_est.txt 2>&1
python3 $R/scratch/bench_shape.py 64 9 2 65536 > $O/shape_64_9_2_65536.txt 2>&1
python3 $R/scratch/bench_shape.py 32 5 2 65536 0.5 > $O/shape_32_5_2_65536.txt 2>&1
python3 $R/scratch/bench_shape.py 128 15 4 65536 > $O/shape_128_15_4_65536.txt 2>&1
python3 $R/scratch/bench_shape.py 256 31 2 8192 0.1 > $O/shape_256_31_2_8192.txt 2>&1
python3 $R/scratch/bench_tx.py > $O/bench_tx.txt 2>&1
python3 $R/scratch/bench_frames.py > $O/bench_frames.txt 2>&1
ls -la $O
