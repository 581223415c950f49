"""A/B of environment settings with one library, interleaved rounds on one device.
Usage: python tools/abenv.py "name1:VAR=VAL,VAR2=VAL" "name2:" [--cfg3]"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
variants = []
extra = []
for a in sys.argv[1:]:
    if a == '--cfg3':
        extra = ['--config', 'cfg3']
        continue
    name, _, envs = a.partition(':')
    env = dict(kv.split('=', 1) for kv in envs.split(',') if kv)
    variants.append((name, env))
res = {n: [] for n, _ in variants}
for r in range(3):
    for name, env in variants:
        e = dict(os.environ, **env)
        out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--steps', '4', '--warmup', '2',
                              '--no-cpu-baseline'] + extra, env=e, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith('{')]
        if not line:
            print(name, 'FAILED', out.stderr[-600:])
            continue
        d = json.loads(line[0])
        k = d['roofline']['kernel_ms_per_step']
        res[name].append((d['ms_per_step'], k['edge_fwd'], k['edge_bwd'], k['col_gather']))
for name, rows in res.items():
    for row in rows:
        print(f'{name:16s} step {row[0]:7.3f}  fwd {row[1]:.3f}  bwd {row[2]:.3f}  col {row[3]:.3f}')
