#!/bin/bash
OUT=gpurun_out/a8; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "log_bin or chebyshev" 2>&1 | tail -12
GRX_BIN_SORT=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "log_bin or chebyshev" 2>&1 | tail -4
for v in 0 1; do
  GRX_BIN_SORT=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-api-wall > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  GRX_BIN_SORT=$v timeout 300 python bench.py --workload er100k --steps 20 --warmup 5 --no-cpu-baseline --no-api-wall > $OUT/er_$v.json 2> $OUT/er_$v.err
  GRX_BIN_SORT=$v timeout 600 python bench.py --workload dw5m --steps 4 --warmup 2 --no-cpu-baseline --no-api-wall > $OUT/dw_$v.json 2> $OUT/dw_$v.err
done
python - <<PY
import json
for w in ("bench","er","dw"):
  for v in (0,1):
    try:
        j=json.loads(open(f"$OUT/{w}_{v}.json").read().strip().splitlines()[-1])
        k=j["kernel_ms_per_step"]
        names=("key_bits_kernel","tile_count_kernel","scan_kernel","scatter_kernel","bin_threshold_kernel","bin_assign_kernel","sel_map_kernel","sel_hist_kernel","sel_walk1_kernel","sel_collect_kernel","sel_sort_kernel","sel_walk2_kernel")
        print(w, "sort" if v else "select", round(j["ms_per_step"],3), round(sum(k.get(x,0) for x in names),3), {x: round(k[x],3) for x in names if k.get(x)})
    except Exception as e:
        print(w, v, "ERR", e, open(f"$OUT/{w}_{v}.err").read()[-500:])
PY
