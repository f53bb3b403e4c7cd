#!/bin/bash
# round-3 check of the sharded path below the ABI: tests, then the bench single vs forced one-rank RCCL
OUT=gpurun_out/a1; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_no_torch_ops.py -x -q -m gpu 2>&1 | grep -v "^\[Gloo\]" | tail -40 > $OUT/tests.log; tail -5 $OUT/tests.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-api-wall > $OUT/bench_single.json 2> $OUT/bench_single.err
GRX_FORCE_COLLECTIVES=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-api-wall > $OUT/bench_forced.json 2> $OUT/bench_forced.err
python - <<PY
import json
for f in ("single","forced"):
    try:
        j=json.loads(open(f"$OUT/bench_{f}.json").read().strip().splitlines()[-1])
        print(f, j["ms_per_step"], j["ms_per_step_without_launch_events"], j["refex"]["ms_per_step"], j["nmf"]["ms_per_step"], (j.get("per_rank") or [{}])[0].get("exchange"))
    except Exception as e:
        print(f, "ERR", e)
PY
