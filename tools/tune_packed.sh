for S in 2 4 8; do
  echo "== GRX_PACKED_SLOTS=$S"
  GRX_PACKED_SLOTS=$S python -m pytest tests/test_gpu_packed.py -q -m gpu -s -k "faster_at_ba1m or power_law or every_output" 2>&1 | grep -E "gen1_packed|passed|failed"
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc off --no-api-wall > gpurun_out/b_ba1m.json 2>/dev/null
python bench.py --workload er100k --steps 20 --warmup 5 --no-cpu-baseline --pmc off --no-api-wall > gpurun_out/b_er.json 2>/dev/null
GRX_NO_PACKED_ROWS=1 python bench.py --workload er100k --steps 20 --warmup 5 --no-cpu-baseline --pmc off --no-api-wall > gpurun_out/b_er_nopack.json 2>/dev/null
python - <<PY
import json
for f in ("b_ba1m","b_er","b_er_nopack"):
    l=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, round(l["ms_per_step"],4), round(l["refex"]["ms_per_step"],4), round(l["nmf"]["ms_per_step"],4), round(l["roofline"]["frac"],4), l["roofline"]["frac_of_gather_ceiling"], round(l["kernel_ms_per_step"]["aggregate_kernel"],4))
PY
