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
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $ctr | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$name -o pmc -- $BENCH > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
done
# keep the merge small (gpurun copies back at most 64 MiB, and the per-dispatch counter rows of a run with 2 000 seeding
# launches per encode are far more): condense HERE, ship the summaries, drop the raw rows
find $OUT -type f ! -name '*.csv' ! -name '*.json' ! -name '*.err' -delete
cd $REPO && python tools/summarize_rocprof.py $TAG $WORKLOAD > $OUT/summarize.log 2>&1
mkdir -p $REPO/gpurun_out/summ_$TAG
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc.json profiles/traffic_latest.json $OUT/binary.json $OUT/summarize.log $REPO/gpurun_out/summ_$TAG/ 2>/dev/null
cp $OUT/trace_bench.json $REPO/gpurun_out/summ_$TAG/ 2>/dev/null
rm -rf $OUT/trace $OUT/pmc_*
du -sh $OUT $REPO/gpurun_out/summ_$TAG
