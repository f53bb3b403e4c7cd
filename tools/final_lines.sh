#!/bin/bash
# frozen binary of a round: timelines, bench lines, forced-collectives lines (run ON THE GPU BOX through gpurun)
OUT=gpurun_out/final; mkdir -p $OUT
export TMPDIR=/tmp
for w in ba1m er100k dw5m; do
  steps=6; [ $w = dw5m ] && steps=3
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$w -o t -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps $steps --warmup 2 --no-cpu-baseline --no-api-wall > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/tl_$w.err )
  f=$(find /tmp/tl_$w -name '*kernel_trace.csv' | head -1)
  python tools/step_timeline.py $f $OUT/timeline_$w.txt
  tail -1 $OUT/timeline_$w.txt
done
timeout 900 python bench.py > $OUT/bench_ba1m.json 2> $OUT/bench_ba1m.err
timeout 900 python bench.py --workload er100k > $OUT/bench_er100k.json 2> $OUT/bench_er100k.err
timeout 1200 python bench.py --workload dw5m --steps 5 --warmup 2 > $OUT/bench_dw5m.json 2> $OUT/bench_dw5m.err
# GRX_FORCE_COLLECTIVES=1 needs the torch.distributed environment: bench.py takes the sharded path only when RANK is set
FORCED="env RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 GRX_FORCE_COLLECTIVES=1"
$FORCED timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-api-wall --no-sharded-extra > $OUT/forced_ba1m.json 2> $OUT/forced_ba1m.err
$FORCED timeout 900 python bench.py --workload dw5m --steps 4 --warmup 2 --no-cpu-baseline --no-api-wall > $OUT/forced_dw5m.json 2> $OUT/forced_dw5m.err
$FORCED timeout 600 python bench.py --workload er100k --steps 20 --warmup 5 --no-cpu-baseline --no-api-wall --no-sharded-extra > $OUT/forced_er100k.json 2> $OUT/forced_er100k.err
python - <<PY
import json
for f in ("bench_ba1m","bench_er100k","bench_dw5m","forced_ba1m","forced_dw5m","forced_er100k"):
    try:
        j=json.loads(open(f"$OUT/{f}.json").read().strip().splitlines()[-1])
        print(f, round(j["ms_per_step"],3), "%.3g"%j["value"], j.get("api_wall_s",{}).get("total_s"), (j.get("cpu_baseline") or {}).get("value"), j["roofline"].get("traffic"), j["roofline"].get("traffic_stale"))
    except Exception as e:
        print(f, "ERR", e, open(f"$OUT/{f}.err").read()[-400:])
PY
