"""Timing-only ablation of the MFMA edge kernels on the cfg2 batch (results are wrong by design).
Each variant runs in a child process: PVS_ABLATE bits: 1 no MFMA, 2 (unused here), 4 no row
reduction, 8 gathers hit 8 hot rows; PVS_EGNN_LIB selects the no-transcendental build."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
VARIANTS = [('baseline', {}), ('no_mfma', {'PVS_ABLATE': '1'}), ('no_reduce', {'PVS_ABLATE': '4'}),
            ('hot_gather', {'PVS_ABLATE': '8'}),
            ('no_silu', {'PVS_EGNN_LIB': str(ROOT / 'pointvs_amd' / 'libpvs_egnn_nosilu.so')}),
            ('no_mfma+no_silu', {'PVS_ABLATE': '1',
                                 'PVS_EGNN_LIB': str(ROOT / 'pointvs_amd' / 'libpvs_egnn_nosilu.so')}),
            ('no_mfma+no_reduce+hot', {'PVS_ABLATE': 'd'}),
            ('all_off', {'PVS_ABLATE': 'd',
                         'PVS_EGNN_LIB': str(ROOT / 'pointvs_amd' / 'libpvs_egnn_nosilu.so')})]

for name, env in VARIANTS:
    e = dict(os.environ, **env)
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--steps', '3', '--warmup', '1',
                          '--no-cpu-baseline'], env=e, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith('{')]
    if not line:
        print(name, 'FAILED', out.stderr[-300:])
        continue
    k = json.loads(line[0])['roofline']['kernel_ms_per_step']
    print(f'{name:24s} edge_fwd {k["edge_fwd"]:.3f}  edge_bwd {k["edge_bwd"]:.3f} ms/step')
