#!/bin/bash
# polling vs blocking waits on the short read-backs
OUT=gpurun_out/b5; mkdir -p $OUT
for v in 0 1 0 1; do
  GRX_BLOCKING_WAITS=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-api-wall > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  GRX_BLOCKING_WAITS=$v timeout 300 python bench.py --workload er100k --steps 30 --warmup 5 --no-cpu-baseline --no-api-wall > $OUT/er_$v.json 2> $OUT/er_$v.err
  python - <<PY
import json
for w in ("bench","er"):
    j=json.loads(open(f"$OUT/{w}_$v.json").read().strip().splitlines()[-1])
    print(w, "blocking" if $v else "polling", round(j["ms_per_step"],3))
PY
done
