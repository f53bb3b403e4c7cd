#!/bin/bash
# kernel timelines of one bench step (ba1m, er100k) and the forced-collectives one-rank step
OUT=gpurun_out/b1; mkdir -p $OUT
export TMPDIR=/tmp
for w in ba1m er100k; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$w -o t -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 6 --warmup 3 --no-cpu-baseline --no-api-wall > $GRAFT_REPO_ROOT/$OUT/bench_$w.json 2> $GRAFT_REPO_ROOT/$OUT/bench_$w.err )
  f=$(find /tmp/tl_$w -name '*kernel_trace.csv' | head -1)
  python tools/step_timeline.py $f $OUT/timeline_$w.txt
done
GRX_FORCE_COLLECTIVES=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-api-wall > $OUT/forced.json 2> $OUT/forced.err
tail -c 600 $OUT/forced.json
