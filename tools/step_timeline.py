#!/usr/bin/env python3
"""Timeline of the last training step in a rocprofv3 --kernel-trace CSV: duration of every launch and the idle gap in
front of it.   usage: tools/step_timeline.py <dir with *_kernel_trace.csv> [anchor kernel substring]"""
import csv
import glob
import sys

rows = None
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    r = list(csv.DictReader(open(f)))
    if rows is None or len(r) > len(rows):
        rows = r
anchor = sys.argv[2] if len(sys.argv) > 2 else 'k_extract_runs'
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
start = idx[-2] if len(idx) > 1 else idx[-1]
stop = idx[-1]
prev_end, busy, gaps = None, 0.0, 0.0
for r in rows[start:stop]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
    prev_end = max(en, prev_end or en)
    busy += (en - st) / 1e3
    gaps += max(gap, 0.0)
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f'{(en - st) / 1e3:9.1f} us  gap {gap:7.1f}  {name[:100]}')
print(f'launches {stop - start}, kernel time {busy / 1e3:.3f} ms, idle between kernels {gaps / 1e3:.3f} ms')
