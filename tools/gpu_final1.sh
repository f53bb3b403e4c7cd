#!/bin/bash
bash tools/profile_gpu.sh r03 ba1m > gpurun_out/prof_r03.log 2>&1
bash tools/profile_gpu.sh r03_dw5m dw5m > gpurun_out/prof_r03_dw5m.log 2>&1
tail -3 gpurun_out/prof_r03.log gpurun_out/prof_r03_dw5m.log
