"""(GPU) Phase timeline of the H = 32 edge backward's tile loop from the -DPVS_TILE_TRACE build
(tools/variant_obj.sh trace edge_bwd_f16.hip "-DPVS_TILE_TRACE"): workgroup 0 stores the shader clock at 16 points of
each of its first 128 tiles per wave; the LAST backward launch of a cfg2 step (layer 0: full work) is what stays in the
buffer. Prints, per phase, the mean cycles a wave spends in it, and how the two waves of a SIMD (w, w + 4) overlap.

Usage: PVS_EGNN_LIB=pointvs_amd/libpvs_egnn_trace.so python tools/tile_trace.py [--config cfg2] [--batch 32]"""
import argparse
import ctypes
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

# trace points in program order (the ids are the PVS_TP arguments) and what lies between a point and the next
ORDER = [21, 0, 1, 18, 19, 20, 2, 3, 4, 16, 17, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]
PHASES = ['graph-boundary walk, next tile\'s index loads issued', 'gather issued -> all of it landed, z1', 'SiLU(z1), SiLU\'(z1) parked', 'a1 scale check', 'a1 split',
          'a1 image write', 'W2 a1 chain -> z2', 'SiLU(z2) -> m', 'gxagg load, m scale check', 'm split', 'm image write',
          'Wc1 m chain, SiLU(zc), g_zc', 'scale + split g_zc, image', 'Wc1^T chain -> g_m',
          'gWc1 product issue, row terms load', 'row terms, g_z2', 'scale + split g_z2, image, W2^T chain',
          'gW2 product issue', 'g_z1, g_rho, records', 'g_z1 tile write', 'row reduction + stores',
          '(loop end -> the previous stores have completed)']
NP = 24


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='cfg2')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--dump', default=None, help='write the raw [8, 128, 16] cycle table to this .npy')
    args = ap.parse_args()
    from pointvs_amd import _lib
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    from pointvs_amd.synthetic import CONFIGS, synthetic_batch
    cfg = CONFIGS[args.config]
    torch.manual_seed(0)
    model = SartorrasEGNN(Path('/tmp/pvs_trace'), 2e-3, 1e-4, silent=True, **cfg['model']).train()
    batch = synthetic_batch(cfg['cfg_id'], args.batch, **cfg['graph']).to('cuda')
    y = batch.y.float()
    for _ in range(2):
        model.optimiser.zero_grad()
        loss = model.get_loss(y, model(batch).reshape(-1))
        loss.backward()
    torch.cuda.synchronize()
    lib = _lib.lib()
    fn = lib.pvs_debug_tile_trace
    fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    fn.restype = ctypes.c_int
    buf = np.zeros((8, 128, NP), dtype=np.uint64)
    rc = fn(buf.ctypes.data, buf.size)
    assert rc == 0, rc
    if args.dump:
        np.save(args.dump, buf)
    t = buf.astype(np.int64)[:, :, ORDER]
    K = len(ORDER)
    # skip the first tiles (cold caches, staggered start) and any unwritten tail
    lo, hi = 8, 120
    seg = t[:, lo:hi, :]
    d_in = np.diff(seg, axis=2)
    d_next = t[:, lo + 1:hi + 1, 0] - seg[:, :, K - 1]         # loop end -> the next tile's first point
    tile = t[:, lo + 1:hi + 1, 0] - seg[:, :, 0]
    print(f'traced kernel: {tile.mean():.0f} cycles per tile per wave (median {np.median(tile):.0f}, '
          f'min {tile.min()}, max {tile.max()}); {hi - lo} tiles x 8 waves of workgroup 0')
    print(f'{"phase":52s} {"mean":>7s} {"median":>7s} {"p90":>7s}  share')
    means = []
    for k in range(K):
        d = d_in[:, :, k] if k < K - 1 else d_next
        means.append(d.mean())
        print(f'{PHASES[k]:52s} {d.mean():7.0f} {np.median(d):7.0f} {np.percentile(d, 90):7.0f}  {100 * d.mean() / tile.mean():5.1f} %')
    # overlap of the two waves of a SIMD: where is wave w + 4 while wave w is in phase k?
    print('\nper SIMD pair (w, w+4): fraction of wave w\'s time in each phase during which its partner is in the SAME phase')
    same = np.zeros(K)
    tot = np.zeros(K)
    for w in range(4):
        a, b = t[w], t[w + 4]
        # partner's phase as a function of time: event list
        bt = b[lo - 4:hi + 4].reshape(-1)
        bph = np.tile(np.arange(K), hi - lo + 8)
        for n in range(lo, hi):
            for k in range(K):
                s0 = a[n, k]
                s1 = a[n, k + 1] if k < K - 1 else a[n + 1, 0]
                if s1 <= s0:
                    continue
                i0 = np.searchsorted(bt, s0, side='right') - 1
                i1 = np.searchsorted(bt, s1, side='left')
                for i in range(max(i0, 0), min(i1, len(bt) - 1)):
                    o = min(s1, bt[i + 1]) - max(s0, bt[i])
                    if o > 0 and bph[i] == k:
                        same[k] += o
                tot[k] += s1 - s0
    for k in range(K):
        print(f'{PHASES[k]:52s} {100 * same[k] / max(tot[k], 1):5.1f} %')


if __name__ == '__main__':
    main()
