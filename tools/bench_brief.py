"""stdin: bench.py's JSON line -> the few figures one looks at between two builds."""
import json
import sys

j = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = j.get('kernel_ms_per_step', {})
pick = {n: round(v, 4) for n, v in k.items() if any(t in n for t in ('tri', 'ego', 'node_info'))}
print(round(j['ms_per_step'], 4), 'kernels', round(sum(k.values()), 4), 'launch-event-free', round(j.get('ms_per_step_without_launch_events', 0), 4), '%.4g' % j['value'], 'refex', round(j['refex']['ms_per_step'], 4), 'nmf', round(j['nmf']['ms_per_step'], 4), pick)
