#!/bin/bash
# counters of the triangle-counting kernel (tools/tri_bench.py under rocprofv3 --pmc, one pass per counter group)
OUT=$GRAFT_REPO_ROOT/gpurun_out/b3; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
for v in 0; do
  python $GRAFT_REPO_ROOT/tools/tri_bench.py 1000000 20 2>&1 | tail -1
  for ctr in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_CYCLES" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_LEVEL_sum TCC_TAG_STALL_sum"; do
    name=$(echo $ctr | tr ' ' '_' | cut -c1-30)
    timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $OUT/v${v}_$name -o pmc -- python $GRAFT_REPO_ROOT/tools/tri_bench.py 1000000 5 > /dev/null 2> $OUT/v${v}_$name.err
  done
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
for v in (0,):
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(f"gpurun_out/b3/v{v}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if 'triangle_count' in r['Kernel_Name']:
                tot[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
    print('variant', v, {k: round(tot[k]/cnt[k]) for k in sorted(tot)})
PY
find $OUT -type f ! -name '*.txt' ! -name '*.err' -delete
