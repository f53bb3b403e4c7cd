#!/bin/bash
OUT=gpurun_out/a5; mkdir -p $OUT
timeout 900 python bench.py --workload dw5m --steps 5 --warmup 2 --no-cpu-baseline > $OUT/dw5m.json 2> $OUT/dw5m.err
GRX_FORCE_COLLECTIVES=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --workload dw5m --steps 5 --warmup 2 --no-cpu-baseline --no-api-wall > $OUT/dw5m_forced.json 2> $OUT/dw5m_forced.err
python - <<PY
import json
for w in ("dw5m","dw5m_forced"):
    try:
        j=json.loads(open(f"$OUT/{w}.json").read().strip().splitlines()[-1])
        print(w, round(j["ms_per_step"],2), round(j["refex"]["ms_per_step"],2), round(j["nmf"]["ms_per_step"],2), j.get("api_wall_s"), (j.get("per_rank") or [{}])[0].get("exchange"))
        print({x: round(v,2) for x,v in list(j["kernel_ms_per_step"].items())[:10]})
    except Exception as e:
        print(w, "ERR", e)
PY
tail -3 $OUT/dw5m.err
