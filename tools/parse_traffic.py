"""Parse the FETCH_SIZE / WRITE_SIZE passes of tools/measure_traffic.sh into a traffic JSON.
Usage: parse_traffic.py <dir with the two pass dirs> <out.json> [bench args...]"""

import csv, glob, json, re, sys
from collections import defaultdict
root, dst = sys.argv[1], sys.argv[2]
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(set))
for f in glob.glob(f'{root}/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r'(k_[A-Za-z0-9_]+)', row['Kernel_Name'])
        if not m: continue
        k = m.group(1)
        agg[k][row['Counter_Name']] += float(row['Counter_Value'])
        cnt[k][row['Counter_Name']].add(row['Dispatch_Id'])
out = {'command': 'rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 bench.py --steps 2 --warmup 1 ' + ' '.join(sys.argv[3:]),
       'note': 'KiB per launch; FETCH_SIZE doubled (gfx950 reports half the bytes of 16-B-per-lane reads, '
               'MI355X_MICROARCH.md HBM section; every bulk read of these kernels is a dwordx4 load); '
               'Infinity-Cache hits are counted', 'kernels': {}}
for k in agg:
    if not any(s in k for s in ('edge_fwd', 'edge_bwd', 'node_gather')): continue
    f = agg[k]['FETCH_SIZE'] / max(len(cnt[k]['FETCH_SIZE']), 1)
    w = agg[k]['WRITE_SIZE'] / max(len(cnt[k]['WRITE_SIZE']), 1)
    out['kernels'][k] = {'fetch_kib_raw': round(f, 1), 'write_kib': round(w, 1),
                         'hbm_bytes_per_launch': int((2 * f + w) * 1024)}
json.dump(out, open(dst, 'w'), indent=1)
print(json.dumps(out['kernels'], indent=1))
