"""Fuzz of the thin operators either side of the layer stack (pointvs_amd/functional.py over the C ABI: pvs_linear_*,
pvs_mean_pool_*, pvs_pool_head_*, pvs_bce_logits_fwd, pvs_segment_reduce_*) against fp64 torch on the CPU, forward AND
backward, on random shapes: row counts 1 ... 70,000, widths 1 ... 300 (the embedding's 12 inputs, the heads' 1 / 16 / 32
outputs, widths that are no multiple of 4 or 32), empty graphs in a batch, segment ids with empty segments and long runs.
1e-5 max(1, max|ref|) per tensor.   usage (GPU box): python tools/fuzz_ops.py [first_seed] [n_seeds]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pointvs_amd import functional as PF  # noqa: E402

TOL = 1e-5


def dist(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    if a.shape != b.shape:
        return float('inf')
    return float((a - b).abs().max() / max(1.0, float(b.abs().max()))) if b.numel() else 0.0


def check(problems, what, got, ref):
    d = dist(got, ref)
    if not d < TOL:
        problems.append(f'{what}: {d:.2e}')
    return d


def run_seed(seed):
    rng = np.random.default_rng(161803 + seed)
    gen = torch.Generator().manual_seed(seed)
    problems, worst = [], 0.0

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=gen, dtype=torch.float64) * scale)

    # ---- linear: y = x W^T + b ----
    n = int(rng.choice([1, 2, 31, 33, 500, 4097, int(rng.integers(1, 70000))]))
    k = int(rng.choice([1, 3, 12, 16, 32, 33, 48, 64, 100, 128, 300]))
    c = int(rng.choice([1, 2, 16, 32, 48, 64, 65, 128]))
    bias = bool(rng.integers(4) > 0)
    x, w, b, gy = rnd(n, k), rnd(c, k, scale=0.3), rnd(c) if bias else None, rnd(n, c)
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if bias else None
    (torch.nn.functional.linear(xr, wr, br) * gy).sum().backward()
    xg, wg = x.float().cuda().requires_grad_(), w.float().cuda().requires_grad_()
    bg = b.float().cuda().requires_grad_() if bias else None
    y = PF.linear(xg, wg, bg)
    (y * gy.float().cuda()).sum().backward()
    tag = f'linear n={n} k={k} c={c} bias={bias}'
    worst = max(worst, check(problems, tag + ' y', y, torch.nn.functional.linear(x, w, b)),
                check(problems, tag + ' g_x', xg.grad, xr.grad), check(problems, tag + ' g_w', wg.grad, wr.grad))
    if bias:
        worst = max(worst, check(problems, tag + ' g_b', bg.grad, br.grad))

    # ---- mean pool and pool + head over contiguous graphs (some of them empty) ----
    n_graphs = int(rng.integers(1, 40))
    counts = rng.integers(0, 400, size=n_graphs)
    counts[rng.integers(0, n_graphs)] += 1          # at least one node in the batch
    ptr = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32)
    nn_ = int(ptr[-1])
    width = int(rng.choice([16, 32, 64, 96, 128]))
    n_out = int(rng.choice([1, 2, 16, 32]))
    h, wh, bh, gp = rnd(nn_, width), rnd(n_out, width, scale=0.3), rnd(n_out), rnd(n_graphs, n_out)
    hr, whr, bhr = h.clone().requires_grad_(), wh.clone().requires_grad_(), bh.clone().requires_grad_()
    seg = torch.repeat_interleave(torch.arange(n_graphs), torch.as_tensor(counts))
    pooled_ref = torch.zeros(n_graphs, width, dtype=torch.float64).index_add(0, seg, hr) / \
        torch.as_tensor(counts, dtype=torch.float64).clamp(min=1).unsqueeze(1)
    out_ref = torch.nn.functional.linear(pooled_ref, whr, bhr)
    (out_ref * gp).sum().backward()
    hg, whg, bhg = (t.float().cuda().requires_grad_() for t in (h, wh, bh))
    out = PF.pool_head(hg, ptr.cuda(), whg, bhg)
    (out * gp.float().cuda()).sum().backward()
    tag = f'pool_head graphs={n_graphs} nodes={nn_} width={width} out={n_out}'
    worst = max(worst, check(problems, tag + ' y', out, out_ref), check(problems, tag + ' g_h', hg.grad, hr.grad),
                check(problems, tag + ' g_w', whg.grad, whr.grad), check(problems, tag + ' g_b', bhg.grad, bhr.grad))
    h2 = h.float().cuda().requires_grad_()
    pooled = PF.mean_pool(h2, ptr.cuda())
    gpool = rnd(n_graphs, width)
    (pooled * gpool.float().cuda()).sum().backward()
    h2r = h.clone().requires_grad_()
    pr = torch.zeros(n_graphs, width, dtype=torch.float64).index_add(0, seg, h2r) / \
        torch.as_tensor(counts, dtype=torch.float64).clamp(min=1).unsqueeze(1)
    (pr * gpool).sum().backward()
    worst = max(worst, check(problems, f'mean_pool graphs={n_graphs} width={width}', pooled, pr),
                check(problems, 'mean_pool g_h', h2.grad, h2r.grad))

    # ---- BCE with logits, mean ----
    nb = int(rng.choice([1, 2, 31, 32, 33, 257, 5000]))
    logit, target = rnd(nb, scale=float(rng.choice([0.1, 3.0, 30.0]))), (torch.rand(nb, generator=gen) > 0.5).double()
    lr = logit.clone().requires_grad_()
    loss_ref = torch.nn.functional.binary_cross_entropy_with_logits(lr, target)
    loss_ref.backward()
    lg = logit.float().cuda().requires_grad_()
    loss = PF.bce_with_logits_mean(lg, target.float().cuda())
    loss.backward()
    worst = max(worst, check(problems, f'bce n={nb} loss', loss, loss_ref), check(problems, f'bce n={nb} grad', lg.grad, lr.grad))

    # ---- unsorted_segment_sum / mean ----
    rows = int(rng.choice([1, 5, 640, 10000, int(rng.integers(1, 200000))]))
    segs = int(rng.choice([1, 3, 100, 5000]))
    wseg = int(rng.choice([1, 3, 4, 32, 33, 64]))
    ids = torch.as_tensor(rng.integers(0, segs, size=rows) if rng.integers(2) else
                          np.sort(rng.integers(0, segs, size=rows)), dtype=torch.int64)
    data, gs = rnd(rows, wseg), rnd(segs, wseg)
    for mean in (False, True):
        dr = data.clone().requires_grad_()
        ref = torch.zeros(segs, wseg, dtype=torch.float64).index_add(0, ids, dr)
        if mean:
            cnt = torch.zeros(segs, dtype=torch.float64).index_add(0, ids, torch.ones(rows, dtype=torch.float64))
            ref = ref / cnt.clamp(min=1).unsqueeze(1)
        (ref * gs).sum().backward()
        dg = data.float().cuda().requires_grad_()
        got = PF.segment_reduce(dg, ids.cuda(), segs, mean=mean)
        (got * gs.float().cuda()).sum().backward()
        tag = f'segment_{"mean" if mean else "sum"} rows={rows} segments={segs} width={wseg}'
        # (a segment of 10^4 ... 10^5 unnormalised terms: the fp32 sum's own rounding exceeds 1e-5 of a result that the
        # terms nearly cancel to - torch's fp32 index_add on the CPU is the noise sample, as in the suite's strict bound)
        ref32 = torch.zeros(segs, wseg).index_add(0, ids, data.float()).double()
        if mean:
            ref32 = ref32 / cnt.clamp(min=1).unsqueeze(1)
        noise = float((ref32 - ref.detach()).abs().max() / max(1.0, float(ref.detach().abs().max())))
        d = dist(got, ref)
        if not d < TOL + 4.0 * noise:
            problems.append(f'{tag}: {d:.2e} (fp32 torch: {noise:.2e})')
        worst = max(worst, min(d, TOL), check(problems, tag + ' grad', dg.grad, dr.grad))
    PF.segment_status_check()
    return worst, problems


if __name__ == '__main__':
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    bad, t0, worst_all = 0, time.time(), (0.0, -1)
    for seed in range(first, first + count):
        try:
            worst, problems = run_seed(seed)
        except Exception as exc:      # noqa: BLE001 - a raising operator is a finding too
            worst, problems = 0.0, [f'raised {type(exc).__name__}: {exc}']
        worst_all = max(worst_all, (worst, seed))
        if problems:
            bad += 1
            print('FAIL', seed, problems[:6], flush=True)
    print(f'done: {count} seeds from {first}, failures: {bad}, worst distance {worst_all[0]:.2e} (seed {worst_all[1]}), '
          f'{time.time() - t0:.0f} s')
