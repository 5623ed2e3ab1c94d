#!/bin/bash
# summarise gpurun_out/ab/*/*_kernel_trace.csv (scratch/ab_run.sh): one line per variant, round, path
for f in gpurun_out/ab/*/*_kernel_trace.csv; do d=$(basename $(dirname $f)); p=$(basename $f _kernel_trace.csv); python3 scratch/trace_by_shape.py $f | python3 -c "
import csv,sys
for r in csv.DictReader(sys.stdin):
    if int(r['launches'])>=20 and 'k_row' in r['kernel'] and not ('modulate' in r['kernel'] and '$p'!='modulate'):
        print('%-16s %-13s %-40s wg=%-6s n=%-4s mean %7s median %7s min %7s'%('$d','$p',r['kernel'],r['workgroups'],r['launches'],r['mean_us'],r['median_us'],r['min_us']))
"; done | sort -k2,2 -k4,4 -k1,1
