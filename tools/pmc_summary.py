"""Aggregate a rocprofv3 --pmc CSV (one row per dispatch and counter) per kernel name.
Usage: python tools/pmc_summary.py <dir with *_counter_collection.csv> [substring ...]"""
import csv
import glob
import re
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    filters = sys.argv[2:]
    files = glob.glob(f'{root}/**/*counter_collection.csv', recursive=True)
    agg = defaultdict(lambda: defaultdict(float))
    count = defaultdict(set)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get('Kernel_Name', '')
                if filters and not any(s in name for s in filters):
                    continue
                mm = re.search(r'(k_[A-Za-z0-9_]+(?:<[^>]*>)?)', name)
                short = mm.group(1) if mm else name[:60]
                agg[short][row['Counter_Name']] += float(row['Counter_Value'])
                count[short].add(row['Dispatch_Id'])
    for k in sorted(agg):
        n = max(len(count[k]), 1)
        print(f'{k}  dispatches={n}')
        for c, v in sorted(agg[k].items()):
            print(f'    {c:34s} total {v:.4g}   per-dispatch {v / n:.4g}')


if __name__ == '__main__':
    main()
