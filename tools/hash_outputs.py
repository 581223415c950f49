#!/usr/bin/env python3
"""Hashes of every traced tensor and gradient of one step per kernel family (tools/soak.py's families), as JSON:
run under two builds of the library (PVS_EGNN_LIB) and diff - a change that claims to be bit for bit neutral must
leave every hash unchanged.   python tools/hash_outputs.py [--families a,b] > hashes.json"""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tools'))
import soak  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--families', default=','.join(soak.FAMILIES))
    ap.add_argument('--graphs', type=int, default=2)
    args = ap.parse_args()
    import torch
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    from pointvs_amd.synthetic import CONFIGS, synthetic_batch
    out = {}
    for fam in args.families.split(','):
        cfg = CONFIGS['cfg2']
        torch.manual_seed(11)
        kw = dict(soak.BASE_KW, **soak.FAMILIES[fam])
        model = SartorrasEGNN(Path('/tmp/pvs_hash'), 2e-3, 1e-4, silent=True, **kw).cuda().train()
        graph_kw = dict(cfg['graph'])
        if kw['k'] >= 64:
            graph_kw['edge_radius'] = 6.0
        batch = synthetic_batch(cfg['cfg_id'], args.graphs, **graph_kw).to('cuda')
        out[fam] = soak.one_repeat(model, batch)
    print(json.dumps(out, indent=0, sort_keys=True))


if __name__ == '__main__':
    main()
