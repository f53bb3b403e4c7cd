#!/bin/bash
OUT=gpurun_out/a7; mkdir -p $OUT
timeout 900 python tools/profile_api.py dw5m 2>&1 | grep -v amdgpu.ids | head -30 > $OUT/profile_dw5m.txt; head -3 $OUT/profile_dw5m.txt
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -12 > $OUT/pytest_full.log; tail -5 $OUT/pytest_full.log
