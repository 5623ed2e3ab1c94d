#!/bin/bash
# configs[1] / configs[2] headline value against warm-up and timed steps (200 steps are a 3 ms region behind 0.3 ms of warm-up)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for c in cfg2 cfg3; do for sw in "200 20" "200 500" "200 2000" "1000 20" "1000 500" "2000 2000"; do set -- $sw
  python3 bench.py --config $c --steps $1 --warmup $2 --no-cpu-baseline --no-paths --no-host-paths --large-batch 0 2>/dev/null > /tmp/line.json
  python3 - $c $1 $2 <<'PY'
import sys, json
d = json.load(open("/tmp/line.json"))
print("%s steps %s warmup %s  value %.1f M  (%.2f us per step)  sustained %.1f M blocks/s" % (sys.argv[1], sys.argv[2], sys.argv[3], d["value"] / 1e6, d["ms_per_step"] * 1e3, d.get("sustained", {}).get("value", 0) / 1e6))
PY
done; done; done
