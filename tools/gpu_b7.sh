#!/bin/bash
OUT=gpurun_out/b7; mkdir -p $OUT
for t in 1 2 4 8; do
  for w in er100k ba1m; do
    GRX_NMF_TILES_PER_WAVE=$t timeout 300 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-api-wall > $OUT/${w}_$t.json 2> $OUT/${w}_$t.err
    python - <<PY
import json
j=json.loads(open("$OUT/${w}_$t.json").read().strip().splitlines()[-1]); k=j["kernel_ms_per_step"]
print("$w tiles/wave $t", round(j["ms_per_step"],3), "w_pass", round(k["nmf_w_pass_kernel"],3), "reduce", round(k["reduce_partials_kernel"],3), "iters", j["nmf"]["iterations_per_step"])
PY
  done
done
timeout 1500 python -m pytest tests/test_gpu_rolx.py tests/test_gpu_sharded.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -4
