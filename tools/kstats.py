"""Per-step breakdown of a rocprofv3 kernel_stats.csv. Usage: kstats.py <csv> <n_steps_incl_warmup>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'kernel time per step: {tot / steps / 1e6:.3f} ms')
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"{float(r['TotalDurationNs']) / steps / 1e6:8.3f} ms/step {int(r['Calls']) / steps:6.1f} calls "
          f"{float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:100]}")
