#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): per-dispatch durations of the k-means++ kernels of ONE run of tools/time_kmeans.py
# -> gpurun_out/prof_kmt_<tag>/durations.txt (one line per kernel name: durations in ns of its dispatches of the last run)
set -u
TAG=${1:-a}
shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_kmt_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python $REPO/tools/time_kmeans.py "$@" > $OUT/log.txt 2> $OUT/trace.err
f=$(find $OUT -name '*kernel_trace.csv' | head -1)
python - "$f" "$OUT/durations.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
out = {}
for r in rows:
    n = r['Kernel_Name']
    key = 'prep' if 'km_prep' in n else 'update' if 'km_update' in n else None
    if key:
        out.setdefault(key, []).append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
with open(sys.argv[2], 'w') as fh:
    for key, v in out.items():
        per = len(v) // 4                      # four runs (one warm-up, three timed): keep the last
        v = v[-per:]
        fh.write(key + ' dur ' + ' '.join(str(e - s) for s, e in v) + '\n')
        fh.write(key + ' start ' + ' '.join(str(s - v[0][0]) for s, e in v) + '\n')
PY
find $OUT -type f ! -name 'durations.txt' ! -name 'log.txt' -delete
tail -2 $OUT/log.txt
