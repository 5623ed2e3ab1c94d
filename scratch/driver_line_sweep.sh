#!/bin/bash
# the driver's own command line (bench.py --gpus 1 --steps 20 --warmup 5): a 20-step burst is a 0.3 ms region.  GFDM_BENCH_EARLY_GC=0: collect between warm-up and timed region (as until round 4)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do for poll in 0 1; do export GFDM_BENCH_EARLY_GC=$poll;
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-paths --no-host-paths --large-batch 0 --sustained-seconds 0 2>/dev/null > /tmp/line.json
  python3 - $poll <<'PY'
import sys, json
d = json.load(open("/tmp/line.json"))
print("early gc %s  steps 20 warmup 5  value %.1f M  (%.2f us per step)" % (sys.argv[1], d["value"] / 1e6, d["ms_per_step"] * 1e3))
PY
done; done
