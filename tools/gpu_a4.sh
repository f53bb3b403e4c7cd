#!/bin/bash
OUT=gpurun_out/a4; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_refex.py tests/test_gpu_rolx.py tests/test_gpu_sharded.py -x -q -m gpu -k "not ba1m" 2>&1 | tail -15 > $OUT/tests.log; tail -4 $OUT/tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python bench.py --workload er100k --steps 20 --warmup 5 --no-cpu-baseline --no-api-wall > $OUT/er100k.json 2> $OUT/er100k.err
python - <<PY
import json
for w in ("bench","er100k"):
    try:
        j=json.loads(open(f"$OUT/{w}.json").read().strip().splitlines()[-1])
        k=j["kernel_ms_per_step"]
        print(w, round(j["ms_per_step"],3), round(j["ms_per_step_without_launch_events"],3), j["roofline"]["frac"], j["roofline"]["avg_launch_ms"], {x: round(v,3) for x,v in list(k.items())[:8]}, j.get("api_wall_s"))
    except Exception as e:
        print(w, "ERR", e)
PY
