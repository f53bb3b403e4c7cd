#!/bin/bash
OUT=gpurun_out/a3; mkdir -p $OUT
for v in nt nont; do
  if [ $v = nont ]; then export GRX_LIB_PATH=$PWD/graphrole_amd/libgrx_nont.so; else unset GRX_LIB_PATH; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-api-wall > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  timeout 600 python bench.py --workload dw5m --steps 5 --warmup 2 --no-cpu-baseline --no-api-wall > $OUT/dw5m_$v.json 2> $OUT/dw5m_$v.err
done
unset GRX_LIB_PATH
python - <<PY
import json
for w in ("bench","dw5m"):
  for v in ("nt","nont"):
    try:
        j=json.loads(open(f"$OUT/{w}_{v}.json").read().strip().splitlines()[-1])
        k=j["kernel_ms_per_step"]
        print(w, v, round(j["ms_per_step"],3), {x: round(k[x],3) for x in ("aggregate_kernel","triangle_count_kernel","aggregate_hub_kernel","egonet_kernel<64>") if x in k})
    except Exception as e:
        print(w, v, "ERR", e)
PY
timeout 900 python -m pytest tests/test_gpu_rolx.py -q -m gpu -k "model_selection_vs_reference" 2>&1 | tail -3
python - <<PY
import json
j=json.load(open("gpurun_out/model_selection.json"))
print({k:v.get("cells_that_differ") for k,v in j.items()})
PY
