"""A/B of library builds in ONE process tree on ONE device (interleaved rounds).
Usage: [AB_ROUNDS=3] [AB_ARGS="--config cfg3"] python tools/ab.py name1=path1.so name2=path2.so"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
libs = [a.split('=', 1) for a in sys.argv[1:] if '=' in a]
rounds = int(os.environ.get("AB_ROUNDS", "3"))
res = {n: [] for n, _ in libs}
for r in range(rounds):
    for name, path in libs:
        e = dict(os.environ, PVS_EGNN_LIB=str(Path(path).resolve()))
        try:
            out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--steps', '4', '--warmup', '2',
                                  '--no-cpu-baseline'] + os.environ.get('AB_ARGS', '').split(), env=e, capture_output=True,
                                 text=True, timeout=float(os.environ.get('AB_TIMEOUT', '240')))
        except subprocess.TimeoutExpired:
            print(name, 'TIMED OUT')
            continue
        line = [l for l in out.stdout.splitlines() if l.startswith('{')]
        if not line:
            print(name, 'FAILED', out.stderr[-400:])
            continue
        d = json.loads(line[0])
        k = d['roofline']['kernel_ms_per_step']
        res[name].append((d['ms_per_step'], k['edge_fwd'], k['edge_bwd'], k['col_gather']))
for name, rows in res.items():
    for row in rows:
        print(f'{name:16s} step {row[0]:7.3f}  fwd {row[1]:.3f}  bwd {row[2]:.3f}  col {row[3]:.3f}')
