#!/bin/bash
# round 6, final binary: everything that goes into profiles/ in one gpurun call (outputs stay under 64 MiB)
OUT=gpurun_out/r06; mkdir -p $OUT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu_full.log 2>&1; tail -3 $OUT/pytest_gpu_full.log
bash tools/profile_gpu.sh r06 ba1m > $OUT/prof_r06.log 2>&1
bash tools/profile_gpu.sh r06_dw5m dw5m > $OUT/prof_r06_dw5m.log 2>&1
bash tools/final_lines.sh > $OUT/final_lines.log 2>&1; tail -8 $OUT/final_lines.log
( for c in 'tools/fuzz_kmeans.py 120 1' 'tools/fuzz_kmeans.py 60 7' 'tools/fuzz_binning.py 200 3' 'tools/fuzz_refex.py 600 2' 'tools/fuzz_rolx.py 200 4'; do echo "== $c"; timeout 900 python $c 2>&1 | grep -v "^  \|Warning" | tail -2; done; echo '== GRX_BIN_BID_MIN_N=0 tools/fuzz_binning.py 150 5'; GRX_BIN_BID_MIN_N=0 timeout 600 python tools/fuzz_binning.py 150 5 2>&1 | tail -1 ) > $OUT/fuzz.log 2>&1; tail -6 $OUT/fuzz.log
bash tools/prof_kmeans.sh f6 6000000 64 > $OUT/pk_f6.txt 2>&1; bash tools/prof_kmeans.sh f30 30000000 512 > $OUT/pk_f30.txt 2>&1
timeout 900 python tools/project_scaling.py ba1m 2> $OUT/proj_ba1m.err | tail -1 > $OUT/projected_scaling_ba1m.json; python tools/show_projection.py $OUT/projected_scaling_ba1m.json | grep '^P'
timeout 300 python tools/time_triangle_slices.py ba1m 8 > $OUT/triangle_slices.txt 2>&1
timeout 600 python bench.py --workload ba10m --steps 3 --warmup 1 --no-cpu-baseline --no-api-wall > $OUT/bench_ba10m.json 2> $OUT/bench_ba10m.err; tail -c 300 $OUT/bench_ba10m.err
du -sh gpurun_out
