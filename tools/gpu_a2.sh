#!/bin/bash
OUT=gpurun_out/a2; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_encode.py tests/test_gpu_handoff.py tests/test_gpu_rolx.py tests/test_gpu_ingest.py -x -q -m gpu 2>&1 | tail -30 > $OUT/tests.log; tail -6 $OUT/tests.log
timeout 300 python tools/time_api.py > $OUT/time_api.log 2>&1; grep -v "^ \|^$" $OUT/time_api.log | head -12
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
j=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["api_wall_s"], j["encode"])
PY
