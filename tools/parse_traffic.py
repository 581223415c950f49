"""Parse the PMC passes of tools/measure_traffic.sh into a traffic JSON.
Usage: parse_traffic.py <dir with the pass dirs> <out.json> [bench args...]"""
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
out = {'command': 'rocprofv3 --kernel-trace --pmc <one pass per counter group> -- python3 bench.py --steps 2 --warmup 1 ' + ' '.join(sys.argv[3:]),
       'note': 'per launch. hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE doubled as '
               'MI355X_MICROARCH.md (HBM section) prescribes for 16-byte-per-lane reads (gfx950 tallies each 128-B '
               'request at 64 B); Infinity-Cache hits are counted. request_bytes_per_launch is the cross-check from '
               'the request counters by size: 32*RDREQ_32B + 64*RDREQ_64B + 128*RDREQ_128B + 64*WRREQ_64B + '
               '32*(WRREQ - WRREQ_64B).', 'kernels': {}}
def per(k, c):
    return agg[k][c] / max(len(cnt[k][c]), 1)
for k in agg:
    if not any(s in k for s in ('edge_fwd', 'edge_bwd', 'node_gather')): continue
    f, w = per(k, 'FETCH_SIZE'), per(k, 'WRITE_SIZE')
    rd = 32 * per(k, 'TCC_EA0_RDREQ_32B_sum') + 64 * per(k, 'TCC_EA0_RDREQ_64B_sum') + 128 * per(k, 'TCC_EA0_RDREQ_128B_sum')
    wr = 64 * per(k, 'TCC_EA0_WRREQ_64B_sum') + 32 * (per(k, 'TCC_EA0_WRREQ_sum') - per(k, 'TCC_EA0_WRREQ_64B_sum'))
    out['kernels'][k] = {'fetch_kib_raw': round(f, 1), 'write_kib': round(w, 1),
                         'hbm_bytes_per_launch': int((2 * f + w) * 1024),
                         'request_read_bytes': int(rd), 'request_write_bytes': int(wr),
                         'request_bytes_per_launch': int(rd + wr)}
json.dump(out, open(dst, 'w'), indent=1)
print(json.dumps(out['kernels'], indent=1))
