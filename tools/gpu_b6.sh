#!/bin/bash
# the bulk download with and without NUMA binding of the copy threads / the pinned ring
python - <<PY
import glob
for f in glob.glob('/sys/bus/pci/devices/*/numa_node'):
    pass
import subprocess, torch
print('numa nodes:', open('/sys/devices/system/node/online').read().strip())
PY
for v in 1 0 1 0; do
  GRX_NUMA_BIND=$v timeout 600 python tools/diag_api_dw5m.py 2>&1 | grep "^rep" | tr '\n' ' '; echo " bind=$v"
done
