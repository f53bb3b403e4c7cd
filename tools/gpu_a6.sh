#!/bin/bash
OUT=gpurun_out/a6; mkdir -p $OUT
GRX_TRI_XCD=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "egonet or gen0 or triangle" 2>&1 | tail -3
for v in 0 1; do
  GRX_TRI_XCD=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-api-wall > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  GRX_TRI_XCD=$v timeout 300 python bench.py --workload er100k --steps 20 --warmup 5 --no-cpu-baseline --no-api-wall > $OUT/er_$v.json 2> $OUT/er_$v.err
done
python - <<PY
import json
for w in ("bench","er"):
  for v in (0,1):
    try:
        j=json.loads(open(f"$OUT/{w}_{v}.json").read().strip().splitlines()[-1])
        k=j["kernel_ms_per_step"]
        print(w, v, round(j["ms_per_step"],3), {x: round(k[x],3) for x in ("triangle_count_kernel","aggregate_kernel") if x in k})
    except Exception as e:
        print(w, v, "ERR", e)
PY
timeout 600 python tools/time_finalize.py dw5m 2>&1 | grep -v amdgpu.ids | tail -8
