#!/bin/bash
# cfg4 / cfg5 headline value against the number of timed steps (the 20-step default is a 10 ms region)
cd $GRAFT_REPO_ROOT
for c in cfg4 cfg5; do for sw in "20 10" "50 10" "100 20" "200 20"; do set -- $sw
  python3 bench.py --config $c --steps $1 --warmup $2 --no-cpu-baseline --no-paths --no-host-paths --large-batch 0 2>/dev/null > /tmp/line.json
  python3 - $c $1 $2 <<'PY'
import sys, json
d = json.load(open("/tmp/line.json"))
print("%s steps %s warmup %s  value %.1f M  (%.1f us per step)  sustained %.1f M blocks/s  kernel %.1f us" % (sys.argv[1], sys.argv[2], sys.argv[3], d["value"] / 1e6, d["ms_per_step"] * 1e3, d.get("sustained", {}).get("value", 0) / 1e6, d["roofline"]["kernel_ms"] * 1e3))
PY
done; done
