#!/bin/bash
# headline loop of bench.py over a list of HIP stream counts (configs[1] and configs[2]), alternating repetitions: burst and sustained M blocks/s
#   bash scratch/streams_sweep.sh "2 3 4 6 8" 2
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${2:-2}); do for s in ${1:-2 3 4 6 8}; do for c in cfg2 cfg3; do
  python3 bench.py --config $c --streams $s --no-cpu-baseline --no-paths --no-host-paths --large-batch 0 2>/dev/null > /tmp/line.json
  python3 - $c $s <<'PY'
import sys, json
d = json.load(open("/tmp/line.json"))
print("%s streams %s  burst %.1f M  sustained %.1f M blocks/s" % (sys.argv[1], sys.argv[2], d["value"] / 1e6, d.get("sustained", {}).get("value", 0) / 1e6))
PY
done; done; done
