#!/bin/bash
# the whole N > 1 path of bench.py on a ONE-GPU box: two and three ranks sharing the GPU, torch.distributed over gloo
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3h; mkdir -p $O
cd $R
for n in 1 2 3; do
  timeout 600 python bench.py --gpus $n --dist-backend gloo --config cfg4 --steps 6 --warmup 2 --sustained-seconds 0 --no-cpu-baseline --ring-mib 1024 > $O/cfg4_n$n.json 2> $O/cfg4_n$n.err; echo "cfg4 n=$n rc=$?"
done
timeout 600 python bench.py --gpus 2 --dist-backend gloo --steps 50 --warmup 5 --sustained-seconds 0 --no-cpu-baseline --ring-mib 512 > $O/cfg2_n2.json 2> $O/cfg2_n2.err; echo "cfg2 n=2 rc=$?"
python - <<PY
import json
for f in ("cfg4_n1","cfg4_n2","cfg4_n3","cfg2_n2"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["n_gpus"], d["scaling"], "%.1f M blocks/s"%(d["value"]/1e6), d["config"]["batch_per_gpu"], d["config"]["blocks_per_step_all_gpus"], [round(v,3) for v in d["output_checksum"]])
    except Exception as e:
        print(f, "FAILED", e); print(open("$O/%s.err"%f).read()[-1500:])
PY
