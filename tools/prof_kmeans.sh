#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel trace of tools/time_kmeans.py -> gpurun_out/prof_km_<tag>/
set -u
TAG=${1:-a}
shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_km_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/tools/time_kmeans.py "$@" > $OUT/log.txt 2> $OUT/trace.err
find $OUT -type f ! -name '*kernel_stats.csv' ! -name '*.txt' ! -name '*.err' -delete
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cut -c1-160 "$f" | head -45; else echo "no kernel_stats.csv"; tail -5 $OUT/trace.err; fi
cat $OUT/log.txt | tail -4
