#!/bin/bash
# re-collect single PMC rows: pmc_missing.sh <tag> "path blocks K M L" ...   (appends to gpurun_out/<tag>/pmc_hbm_traffic_extra.csv)
R=$GRAFT_REPO_ROOT; T=$1; shift; O=$R/gpurun_out/$T; mkdir -p $O/pmcx; cd /tmp; export TMPDIR=/tmp
id=$(python3 -c "import sys; sys.path.insert(0, '$R/gr-gfdm_amd/python'); import gfdm_amd; print(gfdm_amd.build_id())")
for spec in "$@"; do
  set -- $spec; run=$1_$3_$4_$5_$2; reps=40; [ $2 -ge 65536 ] && reps=12; [ $2 -ge 65536 ] && [ $3 -ge 256 ] && reps=6
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmcx/$run/$c -o pmc -- python3 $R/scratch/run_kernel.py $1 $2 $reps 2 $3 $4 $5 > $O/pmcx_$run_$c.log 2>&1 || tail -5 $O/pmcx_$run_$c.log
  done
done
python3 $R/scratch/pmc_summary.py $O/pmcx $id > $O/pmc_hbm_traffic_extra.csv; rm -rf $O/pmcx
cat $O/pmc_hbm_traffic_extra.csv
