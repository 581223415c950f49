"""Fuzz of FusedClipAdam (pointvs_amd/optim.py over pvs_adam_clip_step) against clip_grad_value_ + torch.optim.Adam: random sets of
1-60 parameters of random shapes, 3-14 steps, and everything round 6 made the fast path remember - the SET of parameters with a
gradient changes between steps (None gradients come and go: the plan is rebuilt, counters of late starters differ), gradients
alternate between freshly allocated tensors, views of ONE flat buffer (what the one-call layer stack hands out) and tensors
rewritten in place (the remembered pointer table must be found again, or not, correctly), state_dict round trips in the middle
of a run (a device-mapped load brings the counters in as device tensors), weight decay on and off, clip on and off, a deepcopy
of the optimiser. Parameters, both moments and every step counter must match torch's to 1e-6 after the last step.
usage (GPU box): python tools/fuzz_adam.py [first_seed] [n_seeds]"""
import copy
import io
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pointvs_amd.optim import FusedClipAdam  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(1.0, float(b.abs().max()))) if b.numel() else 0.0


def run_seed(seed):
    rng = np.random.default_rng(99100 + seed)
    gen = torch.Generator().manual_seed(seed)
    n_params = int(rng.integers(1, 61))
    shapes = []
    for _ in range(n_params):
        kind = int(rng.integers(4))
        shapes.append([(1,), (int(rng.integers(1, 70)),), (int(rng.integers(1, 70)), int(rng.integers(1, 140))),
                       (1, int(rng.integers(1, 70)))][kind])
    a = [torch.nn.Parameter(torch.randn(s, generator=gen).cuda()) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    wd = float(rng.choice([0.0, 1e-4, 1e-2]))
    lr = float(rng.choice([2e-3, 1e-1]))
    clip = None if rng.integers(4) == 0 else float(rng.choice([1.0, 0.05]))
    oa = FusedClipAdam(a, lr=lr, weight_decay=wd)
    ob = torch.optim.Adam(b, lr=lr, weight_decay=wd)
    never = set(int(k) for k in rng.choice(n_params, size=int(rng.integers(0, max(1, n_params // 6) + 1)), replace=False))
    late = {int(k): int(rng.integers(1, 5)) for k in rng.choice(n_params, size=int(rng.integers(0, max(1, n_params // 5) + 1)),
                                                                replace=False) if int(k) not in never}
    steps = int(rng.integers(3, 15))
    inplace_bufs = {}
    for step in range(steps):
        mode = int(rng.integers(3))          # 0: fresh tensors, 1: views of one flat buffer, 2: rewritten in place
        flaky = set(int(k) for k in rng.choice(n_params, size=int(rng.integers(0, min(3, n_params + 1))), replace=False)) if rng.integers(3) == 0 else set()
        live = [k for k in range(n_params) if k not in never and late.get(k, 0) <= step and k not in flaky]
        for k in range(n_params):
            a[k].grad = b[k].grad = None
        grads = {k: torch.randn(shapes[k], generator=gen).cuda() * float(rng.choice([0.01, 1.0, 30.0])) for k in live}
        if mode == 1 and live:
            sizes = [a[k].numel() for k in live]
            flat = torch.empty(sum(sizes), device='cuda')
            for k, part in zip(live, flat.split_with_sizes(sizes)):
                part.copy_(grads[k].reshape(-1))
                a[k].grad = part.view(shapes[k])
        elif mode == 2:
            for k in live:
                buf = inplace_bufs.setdefault(k, torch.empty(shapes[k], device='cuda'))
                buf.copy_(grads[k])
                a[k].grad = buf
        else:
            for k in live:
                a[k].grad = grads[k].clone()
        for k in live:
            b[k].grad = grads[k].clone()
        if clip is not None:
            oa.step(clip_value=clip)
            if live:          # (torch's clip refuses an empty gradient list; the fused step takes it: nothing to do)
                torch.nn.utils.clip_grad_value_(b, clip)
        else:
            oa.step()
        ob.step()
        if rng.integers(5) == 0:             # checkpoint round trip, sometimes device-mapped (the counters arrive as device tensors)
            blob = io.BytesIO()
            torch.save(oa.state_dict(), blob)
            blob.seek(0)
            sd = torch.load(blob, map_location='cuda' if rng.integers(2) else None, weights_only=False)
            oa.load_state_dict(sd)
        if rng.integers(8) == 0:             # the run continues on a copy of parameters and optimiser
            pair = copy.deepcopy((a, oa))
            a, oa = pair
    problems = []
    worst = 0.0
    for k, (pa, pb) in enumerate(zip(a, b)):
        d = rel(pa, pb)
        worst = max(worst, d)
        if not d < 1e-6:
            problems.append(f'parameter {k} {tuple(shapes[k])}: {d:.2e}')
    sa, sb = oa.state_dict()['state'], ob.state_dict()['state']
    if set(sa) != set(sb):
        problems.append(f'state for {sorted(set(sa) ^ set(sb))} on one side only')
    for k in set(sa) & set(sb):
        if float(sa[k]['step']) != float(sb[k]['step']):
            problems.append(f'step counter of {k}: {float(sa[k]["step"])} vs {float(sb[k]["step"])}')
        if sa[k]['step'].is_cuda:
            problems.append(f'step counter of {k} left on the device')
        for name in ('exp_avg', 'exp_avg_sq'):
            d = rel(sa[k][name], sb[k][name])
            worst = max(worst, d)
            if not d < 1e-6:
                problems.append(f'{name} of {k}: {d:.2e}')
    return (n_params, steps, wd, clip), worst, problems


if __name__ == '__main__':
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    bad, t0, worst_all = 0, time.time(), (0.0, -1)
    for seed in range(first, first + count):
        shape, worst, problems = run_seed(seed)
        worst_all = max(worst_all, (worst, seed))
        if problems:
            bad += 1
            print('FAIL', seed, problems[:5], '(parameters, steps, weight decay, clip) =', shape, flush=True)
    print(f'done: {count} seeds from {first}, failures: {bad}, worst distance {worst_all[0]:.2e} (seed {worst_all[1]}), '
          f'{time.time() - t0:.0f} s')
