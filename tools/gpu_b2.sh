#!/bin/bash
# A/B of the two triangle-counting kernels
OUT=gpurun_out/b2; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_refex.py tests/test_gpu_ingest.py tests/test_gpu_sharded.py -q -m gpu -x -k "triangle or egonet or ego or orient or ingest" 2>&1 | tail -4
for v in 0 1; do
  GRX_TRIANGLES_SHUFFLE=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-api-wall > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  GRX_TRIANGLES_SHUFFLE=$v timeout 300 python bench.py --workload er100k --steps 20 --warmup 5 --no-cpu-baseline --no-api-wall > $OUT/er_$v.json 2> $OUT/er_$v.err
done
python - <<PY
import json
for w in ("bench","er"):
  for v in (0,1):
    try:
        j=json.loads(open(f"$OUT/{w}_{v}.json").read().strip().splitlines()[-1])
        k=j["kernel_ms_per_step"]
        print(w, "shuffle" if v else "regs", round(j["ms_per_step"],3), round(k.get("triangle_count_kernel",0),4))
    except Exception as e:
        print(w, v, "ERR", e, open(f"$OUT/{w}_{v}.err").read()[-500:])
PY
