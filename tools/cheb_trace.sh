cd /tmp; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --workload dw5m --steps 2 --warmup 1 --no-cpu-baseline --no-api-wall --pmc off --soak-seconds 0 > /dev/null 2>&1; python - <<PY
import csv,glob
f=glob.glob("/tmp/tr/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
gk=[k for k in rows[0].keys() if "Grid" in k][0]
ch=[((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r[gk]) for r in rows if "chebyshev" in r["Kernel_Name"]]
print([ (round(a),g) for a,g in ch[-27:]])
PY
