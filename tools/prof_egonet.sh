#!/bin/bash
# ON THE GPU BOX: kernel trace + counter passes of tools/ab_egonet.py (general ego-net path, config-5 graph)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_egonet
mkdir -p $OUT
python $REPO/tools/ab_egonet.py ${1:-dw5m} > /dev/null 2>&1      # publishes the graph in /dev/shm
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/ab_egonet.py ${1:-dw5m}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.json 2> $OUT/trace.err
for ctr in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" FETCH_SIZE "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_TA_BUSY_sum"; do
  name=$(echo $ctr | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
done
find $OUT -type f ! -name '*.csv' ! -name '*.json' ! -name '*.err' -delete
python - <<PY
import csv, glob, collections, re, json
src = '$OUT'
def short(n): return re.sub(r'\(.*$', '', re.sub(r'^void ', '', re.sub(r'\(anonymous namespace\)::', '', n)))
rows = collections.defaultdict(lambda: [0, 0.0])
for p in glob.glob(src + '/trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        k = short(r['Kernel_Name']); rows[k][0] += 1; rows[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print('kernel,calls,avg_us')
for k, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:12]: print(f'{k},{c},{t / c:.1f}')
pmc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for p in glob.glob(src + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        c = pmc[short(r['Kernel_Name'])][r['Counter_Name']]; c[0] += 1; c[1] += float(r['Counter_Value'])
for k in pmc:
    if 'egonet' in k: print(k, json.dumps({c: round(v[1] / v[0], 1) for c, v in sorted(pmc[k].items())}))
PY
