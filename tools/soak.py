#!/usr/bin/env python3
"""Soak run: is every kernel family of the path run-to-run reproducible, bit for bit - also when the
memory it allocates holds stale bytes, and beside another process that loads the GPU? (GPU box.)

    python tools/soak.py --repeats 300 --out gpurun_out/soak.json [--load] [--families default,h64_att,...]

Round 2 saw, once, on one leased box, 1-2 ulp run-to-run differences in the LOGITS of cfg2-shaped
batches that no later run reproduced. Bit-level differences between runs of the
same inputs can only come from (a) an order-dependent reduction (atomics, an unordered LDS hand-off),
(b) a read of memory that this step has not written (then the result depends on what the allocator
hands out: zeros on a fresh box, the previous step's bytes later), or (c) the machine. This tool
separates them. Every repeat hashes the per-layer node features and coordinates, the logits, the loss
and EVERY parameter gradient (not only the logits), in three allocator states:
  clean    torch's caching allocator as it is (a block usually comes back with the previous repeat's bytes)
  nan      every cached block overwritten with NaN before the repeat: a read-before-write of a value that
           matters turns the hashes into NaN hashes
  garbage  every cached block overwritten with finite random bits: such a read changes the hashes
and optionally (--load) beside a second process that keeps the GPU busy with its own kernels.
A family passes when all repeats of all states give ONE hash per tensor.
"""
import argparse
import hashlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

BASE_KW = dict(dim_input=12, k=32, dim_output=1, num_layers=3, residual=False, edge_residual=False,
               edge_attention=False, normalize=False, tanh=False, dropout=0.0, graphnorm=False,
               update_coords=True, permutation_invariance=False, node_attention=False,
               gated_residual=False, rezero=False, softmax_attention=False, model_task='classification')

# one entry per kernel family / instantiation that a model can dispatch
FAMILIES = {
    'default': dict(),                                                     # k_edge_bwd_f16<0,false>
    'h32_att': dict(edge_attention=True, node_attention=True, residual=True),     # <0,true>
    'h32_edgeres': dict(edge_residual=True, tanh=True),                    # <1,false>
    'h32_edgeres_att': dict(edge_residual=True, edge_attention=True),
    'h32_edgeres_rezero': dict(edge_residual=True, residual=True, rezero=True),         # <2,false>
    'h32_edgeres_gated': dict(edge_residual=True, residual=True, gated_residual=True),  # the gated kind without attention
    'h32_edgeres_gated_att': dict(edge_residual=True, residual=True, gated_residual=True, edge_attention=True),   # <3,true>
    # (golden case c3_all_on_k32_g5's flag set: it failed ONCE in eight runs of the whole GPU suite in round 6 and never alone)
    'h32_all_on': dict(num_layers=3, residual=True, gated_residual=True, edge_residual=True, edge_attention=True,
                       node_attention=True, normalize=True, tanh=True, graphnorm=True),
    'h32_softmax_gn': dict(edge_attention=True, softmax_attention=True, graphnorm=True, node_attention=True,
                           residual=True),
    'h64': dict(k=64),                                                     # k_edge_bwd_h64<0,false>
    'h64_att': dict(k=64, edge_attention=True, node_attention=True),       # cfg3's layers
    'h64_edgeres_att': dict(k=64, edge_residual=True, edge_attention=True, tanh=True),
    'generic_h16': dict(k=16, normalize=True),                             # edge_v0 kernels
    'wide128': dict(k=128),                                                # k_edge_bwd_wide<4,0,false>, two-launch forward
    'wide96_edgeres_att': dict(k=96, edge_residual=True, edge_attention=True, node_attention=True, tanh=True),
}


def _hash(t):
    return hashlib.blake2b(t.detach().contiguous().cpu().numpy().tobytes(), digest_size=8).hexdigest()


def poison(mode, reserve_bytes):
    """Overwrite what the caching allocator will hand out next: one large block (served to the big
    per-edge buffers) and a crowd of small ones (the small-allocation pool)."""
    import torch
    if mode == 'clean':
        return
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    blocks = [torch.empty(reserve_bytes // 4, dtype=torch.float32, device='cuda')]
    blocks += [torch.empty(64 * 1024, dtype=torch.float32, device='cuda') for _ in range(512)]     # 256 KiB each
    blocks += [torch.empty(4 * 1024 * 1024, dtype=torch.float32, device='cuda') for _ in range(64)]  # 16 MiB each
    for b in blocks:
        if mode == 'nan':
            b.fill_(float('nan'))
        else:       # finite garbage of every magnitude
            b.view(torch.int32).random_(0, 0x7F000000)
    torch.cuda.synchronize()
    del blocks


def one_repeat(model, batch, trace_layers=True):
    """One training-mode forward + backward; returns {name: hash}."""
    import copy
    import torch
    from pointvs_amd.graph import prepared_for, runs_layout
    g = copy.copy(batch)
    g.__dict__ = dict(batch.__dict__)
    out = {}
    model.zero_grad()
    if trace_layers:
        with torch.no_grad():
            feats, edges, coords, eattr, _ = model.unpack_graph(g)
            pg = prepared_for(edges, eattr, feats.size(0), layout=runs_layout(g, feats.device))
            trace = {}
            model.embed_prepared(pg, feats, coords, trace=trace)
            for k, v in trace.items():
                out[k] = _hash(v)
    y = model(g).reshape(-1)
    loss = model.get_loss(torch.ones_like(y), y)
    loss.backward()
    out['logits'], out['loss'] = _hash(y), _hash(loss)
    for n, p in model.named_parameters():
        if p.grad is not None:
            out['grad/' + n] = _hash(p.grad)
    return out


def soak_family(name, changes, repeats, n_graphs, states=('clean', 'nan', 'garbage'), seed=11, log=print):
    import torch
    from pointvs_amd import graph as pgraph
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    from pointvs_amd.synthetic import CONFIGS, synthetic_batch
    cfg = CONFIGS['cfg2']
    torch.manual_seed(seed)
    kw = dict(BASE_KW, **changes)
    model = SartorrasEGNN(Path('/tmp/pvs_soak'), 2e-3, 1e-4, silent=True, **kw).cuda().train()
    graph_kw = dict(cfg['graph'])
    if kw['k'] >= 64:
        graph_kw['edge_radius'] = 6.0          # cfg3's sparser graphs
    batch = synthetic_batch(cfg['cfg_id'], n_graphs, **graph_kw).to('cuda')
    cache = pgraph.CACHE_ENABLED
    pgraph.CACHE_ENABLED = False               # every repeat prepares its graph afresh, like a training step
    try:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()   # (the poison covers THIS family's peak, not an earlier one's)
        one_repeat(model, batch)               # warm-up (lazy allocations, module loading)
        torch.cuda.synchronize()
        reserve = int(torch.cuda.max_memory_allocated() * 1.25) + (256 << 20)
        seen, first_bad = {}, None
        t0 = time.time()
        for state in states:
            for r in range(repeats):
                poison(state, reserve)
                h = one_repeat(model, batch)
                for key, val in h.items():
                    s = seen.setdefault(key, {})
                    s[val] = s.get(val, 0) + 1
                    if len(s) > 1 and first_bad is None:
                        first_bad = dict(state=state, repeat=r, tensor=key)
        bad = sorted(k for k, s in seen.items() if len(s) > 1)
        rec = dict(family=name, flags=changes, graphs=n_graphs, nodes=int(batch.x.shape[0]),
                   edges=int(batch.edge_index.shape[1]), repeats_per_state=repeats, states=list(states),
                   tensors_hashed=len(seen), tensors_with_more_than_one_hash=bad, first_difference=first_bad,
                   seconds=round(time.time() - t0, 1))
        if bad:
            rec['hash_counts'] = {k: seen[k] for k in bad[:8]}
        log(json.dumps(rec))
        return rec
    finally:
        pgraph.CACHE_ENABLED = cache


def _load_process(stop_path):
    """Second process on the same GPU: back-to-back GEMMs and a model step of its own until told to stop."""
    import torch
    torch.cuda.set_device(0)
    a = torch.randn(4096, 4096, device='cuda')
    b = torch.randn(4096, 4096, device='cuda')
    big = torch.empty(64 << 20, dtype=torch.float32, device='cuda')
    while not Path(stop_path).exists():
        for _ in range(20):
            a = (a @ b).tanh_()
            big.normal_()
        torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--repeats', type=int, default=300, help='per allocator state')
    ap.add_argument('--graphs', type=int, default=4, help='cfg2-shaped graphs per batch')
    ap.add_argument('--families', default=','.join(FAMILIES))
    ap.add_argument('--states', default='clean,nan,garbage')
    ap.add_argument('--load', action='store_true', help='run beside a second process that loads the GPU')
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    loader, stop = None, None
    if args.load:        # started from the fork server BEFORE this process touches the GPU
        import multiprocessing as mp
        import tempfile
        ctx = mp.get_context('forkserver')
        stop = Path(tempfile.mkdtemp()) / 'stop'
        loader = ctx.Process(target=_load_process, args=(str(stop),), daemon=True)
        loader.start()
        time.sleep(5)
    import torch
    results = []
    try:
        for fam in args.families.split(','):
            results.append(soak_family(fam, FAMILIES[fam], args.repeats, args.graphs,
                                       tuple(args.states.split(','))))
    finally:
        if loader is not None:
            stop.write_text('')
            loader.join(timeout=60)
    summary = dict(device=torch.cuda.get_device_name(0), concurrent_gpu_load=bool(args.load),
                   all_reproducible=all(not r['tensors_with_more_than_one_hash'] for r in results),
                   families=results)
    text = json.dumps(summary, indent=1)
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(text + '\n')
    print('ALL REPRODUCIBLE' if summary['all_reproducible'] else 'DIFFERENCES FOUND')
    return 0 if summary['all_reproducible'] else 1


if __name__ == '__main__':
    sys.exit(main())
