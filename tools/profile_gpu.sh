#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py.
# Outputs go to gpurun_out/prof_<tag>/ ; tools/summarize_rocprof.py condenses them into profiles/.
set -u
TAG=${1:-r04}
WORKLOAD=${2:-ba1m}
case "$TAG" in -*) echo "usage: tools/profile_gpu.sh <tag> [workload]  (a tag must not start with '-')"; exit 2;; esac
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
# what was profiled: bench.py refuses to attach these counters to a run of another binary (traffic_stale)
python - <<PY > $OUT/binary.json
import hashlib, json
h = lambda p: hashlib.sha256(open(p, 'rb').read()).hexdigest()
print(json.dumps({'lib_sha256': h('$REPO/graphrole_amd/libgrx.so'), 'bench_sha256': h('$REPO/bench.py'), 'workload': '$WORKLOAD'}))
PY
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --workload $WORKLOAD --steps 3 --warmup 1 --no-cpu-baseline --no-api-wall --soak-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace_bench.json 2> $OUT/trace.err
# one --pmc pass per counter group (never combined with a trace domain); the last two groups are the
# matrix-core counters north_star asks for: fp64 MFMA ops, MFMA busy cycles, and the cycle base they
# are divided by (tools/summarize_rocprof.py: mfma_util)
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE"; do
  name=$(echo $ctr | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$name -o pmc -- $BENCH > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
done
# keep the merge small: drop anything that is not a csv / json / err
find $OUT -type f ! -name '*.csv' ! -name '*.json' ! -name '*.err' -delete
du -sh $OUT
ls -R $OUT | head -50
