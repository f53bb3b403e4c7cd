"""Kernel timeline of the LAST bench step from a rocprofv3 --kernel-trace csv: start (ms), idle gap before the
kernel, duration, name; the step ends before the first k-means / quantisation kernel (the bench's untimed extras).
Usage: python tools/step_timeline.py <kernel_trace.csv> <out.txt> [first-kernel-substring]"""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
anchor = sys.argv[3] if len(sys.argv) > 3 else 'row_sums_kernel'
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the step starts with the generation-0 degree kernels: take the last run of them
idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
start = idx[-1]
while start - 1 in idx:
    start -= 1
t0 = prev_end = int(rows[start]['Start_Timestamp'])
with open(dst, 'w') as out:
    total_gap = total_dur = 0.0
    for r in rows[start:]:
        if any(tag in r['Kernel_Name'] for tag in ('km_', 'q_', 'tile_count_kernel<true>')):
            break
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        total_gap += (s - prev_end) / 1e6
        total_dur += (e - s) / 1e6
        out.write('%9.3f gap %7.3f dur %8.3f  %s\n' % ((s - t0) / 1e6, (s - prev_end) / 1e6, (e - s) / 1e6, r['Kernel_Name'][:100]))
        prev_end = e
    out.write('# step: %.3f ms of kernels and copies, %.3f ms idle between them (host waits and launch gaps)\n' % (total_dur, total_gap))
