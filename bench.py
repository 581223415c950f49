#!/usr/bin/env python3
"""Benchmark of the EGNN hot path: protein-ligand graphs/s, forward + backward + optimiser step.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md §8d "cfg2"): 3-layer EGNN, 32 channels, CLI-default
layer flags, batch of 32 synthetic point clouds of 2000 atoms, edge radius 10 A
(~3.2e5 directed edges per graph), fp32. A step = graph preparation (COO -> CSR/CSC, every step,
as a fresh batch would need) + forward + BCE loss + backward + gradient all-reduce (N>1) +
clip_grad_value_(1.0) + Adam, the reference's `unpack_input_data_and_predict` + `backprop`
(point_neural_network_base.py:176-199, 417-429) without the per-step host read of the loss.
Inputs are resident in HBM before the timed region. One rank per GPU; every rank holds its own
32 graphs (weak scaling), graphs never cross ranks, the only collective is the gradient all-reduce.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import faulthandler  # noqa: E402

import numpy as np  # noqa: E402

faulthandler.enable()
# torch is imported by main() AFTER the self-launch decision: the launcher parent of a multi-rank run never
# maps libamdhip64 / libtorch_hip, let alone initialises HIP (a process that has must not start ranks)
torch = None
dist = None


def _import_torch():
    global torch, dist
    import torch as _torch
    import torch.distributed as _dist
    torch, dist = _torch, _dist

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3   # fp32 vector = fp32 MFMA peak


# what "dtype": "f32" means on this path: inputs, outputs, every accumulation and all
# elementwise work are fp32; the per-edge matrix products run on the 16-bit matrix pipes with each fp32 operand split
# into parts - two fp16 parts under a power-of-two scale and three partial products (H = 32 forward and backward, H = 64
# forward: O(2^-22 |a||b|), between a sequential fp32 FMA chain and bf16x3: tools/f16x2_numerics.py), or three bf16
# parts and six partial products (H = 64 backward: O(2^-25 |a||b|)) - accumulated in fp32; parity at 1e-5 relative
# against the fp64 oracle at BASELINE size is part of the GPU suite
ARITHMETIC = ('fp32 I/O, accumulation and elementwise work; no reduced-precision storage. Per-edge matrix products on the '
              'matrix cores with fp32 accumulation: H=32 forward and backward and H=64 forward as 3-term fp16 products '
              '(two fp16 parts per operand = 22 bits, power-of-two operand scales: per EDGE in the forward, per 32-edge '
              'tile in the backward - without edge attention a scale kept while the tile maximum stays within two binades of its ceiling - where '
              'an element 2^-k below its tile\'s largest keeps 22 - max(0, k - 14) bits: absolute error 2^-36 of the '
              'tile maximum (k - 16 / 2^-38 with edge attention: per-tile scales)), H=64 backward as 6-term bf16 products (three bf16 parts = 24 '
              'bits, no scales); node-level products (N << E) on exact fp32 MFMAs')


def _kernel_ms(lib, name):
    """(summed milliseconds, launches) of one profiled kernel group (pvs_profile_read)."""
    tot, cnt = C.c_double(0.0), C.c_int64(0)
    rc = lib.pvs_profile_read(name.encode(), C.byref(tot), C.byref(cnt))
    return (tot.value, cnt.value) if rc == 0 else (0.0, 0)


def _kernel_each(lib, name, cap=4096):
    """Per-launch milliseconds of one profiled kernel group, in launch order (pvs_profile_read_each)."""
    buf, cnt = (C.c_double * cap)(), C.c_int64(0)
    rc = lib.pvs_profile_read_each(name.encode(), buf, cap, C.byref(cnt))
    return [buf[k] for k in range(min(cnt.value, cap))] if rc == 0 else []


def scaling_note(args, world, strong):
    """How this line relates to BASELINE config 4 (batch 256 over 8 GPUs)."""
    if world == 1:
        return None
    if strong:
        return (f'strong scaling: global batch {args.global_batch} fixed, {args.batch} graphs per GPU at '
                f'{world} GPUs (BASELINE config 4 = --global-batch 256)')
    return (f'weak scaling: {args.batch} graphs per GPU, global batch {world * args.batch}; the strong-scaling '
            f'form of BASELINE config 4 is `--global-batch 256` ({256 // world} graphs per GPU at {world} GPUs), '
            f'and at 8 GPUs the two coincide (8 x 32 = 256)')


def algorithmic_bytes_per_layer(n, e, h):
    """SURVEY.md §8d "store-m" formula: fwd + bwd HBM bytes of one layer on one batch."""
    return 8 * e * h + 14 * e + 6 * (4 * h + 12) * n


def algorithmic_bytes_edge_bwd(n, e, h):
    """Backward share of that formula = what one edge-backward launch must move: read m (4EH),
    read col + type + reverse index (9E), node-level reads/writes (3(4H+12)N)."""
    return 4 * e * h + 9 * e + 3 * (4 * h + 12) * n


def algorithmic_flops_per_step(n, e, h, a, layers, edge_att=False, node_att=False, training=True):
    """SURVEY.md §8d reference-formulation FLOPs (2/MAC, GEMM-like terms of the reference's own
    formulation): forward, x3 for a training step (fwd + dgrad + wgrad)."""
    f_e = 2 * h * (2 * h + 1 + a) + 2 * h * h + (2 * h * h + 2 * h) + (2 * h if edge_att else 0)
    f_n = 6 * h * h + (2 * h if node_att else 0)
    return (3 if training else 1) * layers * (f_e * e + f_n * n)


def executed_flops_per_step(n, e, h, layers, edge_att=False, node_att=False, training=True):
    """FLOPs the kernels actually execute. The per-node split of edge_mlp.0 removes the
    [E, 2H+4] x [2H+4, H] product: per edge the forward runs W2 and Wc1 (4H^2 + 2H for wc2, + 2H for the
    attention logit), the backward recomputes those two and runs two dgrad and two wgrad products
    (12H^2 + 4H); per node the forward runs P, Q (4H^2) and the node MLP (6H^2), the backward their
    dgrad and wgrad (20H^2)."""
    att = 2 * h if edge_att else 0
    f_e_fwd = 4 * h * h + 2 * h + att
    f_n_fwd = 10 * h * h + (2 * h if node_att else 0)
    if not training:
        return layers * (f_e_fwd * e + f_n_fwd * n)
    f_e_bwd = 12 * h * h + 4 * h + 2 * att
    f_n_bwd = 20 * h * h + (4 * h if node_att else 0)
    return layers * ((f_e_fwd + f_e_bwd) * e + (f_n_fwd + f_n_bwd) * n)


def algorithmic_bytes_edge_fwd(n, e, h):
    """Forward share of the §8d formula when the messages are not materialised (inference, cfg5:
    `5E + 2(4H+12)N` per layer): col + type per edge, node rows in and out."""
    return 5 * e + 2 * (4 * h + 12) * n


def flag_summary(model_kwargs):
    on = [k for k in ('residual', 'edge_residual', 'edge_attention', 'node_attention', 'normalize', 'tanh',
                      'graphnorm', 'permutation_invariance', 'gated_residual', 'rezero', 'softmax_attention')
          if model_kwargs.get(k)]
    return 'CLI-default layer flags (all off)' if not on else 'layer flags on: ' + ', '.join(on)


def measured_traffic(config, kernel):
    """HBM bytes per launch of `kernel` from the PMC passes of the same command (separate rocprofv3
    --pmc runs, tools/measure_traffic.sh -> profiles/rNN_<config>_traffic.json; a PMC pass cannot run
    inside this process). The newest round's file wins; None when the config was never measured."""
    for tfile in sorted((ROOT / 'profiles').glob(f'r[0-9][0-9]_{config}_traffic.json'), reverse=True):
        kern = json.loads(tfile.read_text()).get('kernels', {})
        for name, rec in kern.items():
            if name.startswith(kernel) and rec.get('hbm_bytes_per_launch'):
                return rec['hbm_bytes_per_launch'], tfile.name
    return None, None


def measured_limiter(config, kernel):
    """What the dominant kernel waits for, from the newest SQ counter summary of the same command
    (tools/pmc_sq.sh + tools/pmc_summary.py -> profiles/rNN_<config>_pmc_sq.txt; like `traffic`, a PMC pass cannot run
    inside this process): VALU busy and matrix-pipe busy as fractions of SIMD time, the share of a wave's cycles spent
    in s_waitcnt. `bound: "hbm"` and `frac` keep SURVEY 8d's definition; these fields say what actually binds.
    `limiter` is DERIVED from the parsed shares: every resource above its threshold (VALU or matrix pipe
    busy >= 50 % of SIMD time, a wave parked in s_waitcnt >= 30 % of its cycles), else the largest share; the shares are
    those of the build the file was measured on (`limiter_source`, `limiter_commit` when the file records one), not of
    this run."""
    import re
    for pfile in sorted((ROOT / 'profiles').glob(f'r[0-9][0-9]_{config}_pmc_sq.txt'), reverse=True):
        text = pfile.read_text()
        commit = re.search(r'^commit:\s*(\S+)', text, re.M)
        for line in text.splitlines():
            if line.strip().startswith(kernel) and 'VALU busy' in line:
                m = re.search(r'VALU busy (\d+)% of SIMD time, matrix pipe (\d+)%.*?issuing (\d+)%, issue-stalled (\d+)%, '
                              r'in s_waitcnt (\d+)%', line)
                if m:
                    valu, mat, issuing, stalled, wait = (int(g) / 100.0 for g in m.groups())
                    shares = {'valu_issue': valu, 'matrix_pipe': mat, 'wave_stalls': wait}
                    over = [k for k, lim in (('valu_issue', 0.5), ('matrix_pipe', 0.5), ('wave_stalls', 0.3)) if shares[k] >= lim]
                    label = '+'.join(over) if over else max(shares, key=shares.get)
                    return {'limiter': label, 'valu_busy': valu, 'matrix_busy': mat,
                            'waitcnt_share': wait, 'issue_stalled_share': stalled, 'limiter_source': pfile.name,
                            'limiter_commit': commit.group(1) if commit else None}
    return {'limiter': None, 'valu_busy': None, 'matrix_busy': None, 'waitcnt_share': None, 'limiter_source': None,
            'limiter_commit': None}


def visible_gpu_count(topology='/sys/class/kfd/kfd/topology/nodes'):
    """GPUs this process's children would see, WITHOUT loading or initialising the HIP runtime: the KFD
    topology lists one node per agent, GPUs are the nodes with SIMDs; HIP_/ROCR_/CUDA_VISIBLE_DEVICES
    narrow that list. None when the topology is not readable (no driver in this container)."""
    nodes = sorted(Path(topology).glob('*/properties'),
                   key=lambda p: int(p.parent.name))
    if not nodes:
        return None
    count = 0
    for props in nodes:
        try:
            fields = dict(line.split()[:2] for line in props.read_text().splitlines() if len(line.split()) >= 2)
        except OSError:
            continue
        count += int(fields.get('simd_count', '0')) > 0
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        listed = os.environ.get(var)
        if listed is not None:
            count = min(count, len([x for x in listed.split(',') if x.strip()]))
    return count


RCCL_ENV_KEYS = ('NCCL_PROTO', 'NCCL_ALGO', 'NCCL_MIN_NCHANNELS', 'NCCL_MAX_NCHANNELS', 'NCCL_P2P_LEVEL', 'NCCL_DEBUG',
                 'RCCL_MSCCL_ENABLE', 'RCCL_MSCCLPP_ENABLE', 'RCCL_MSCCL_FORCE_ENABLE', 'HSA_ENABLE_IPC_MODE_LEGACY')


def apply_rccl_choice(args, env):
    """--rccl-proto / --rccl-algo -> NCCL_PROTO / NCCL_ALGO in `env`, BEFORE any process group exists (RCCL reads them
    when the communicator is created). The gradient exchange is one 92 KB (cfg2 / cfg4) or 1.42 MB (cfg3) fp32
    all-reduce per step: latency-bound, the regime the LL / LL128 protocols and the tree algorithm exist for, while RCCL's
    own tuning picks by message size and topology. Nothing is forced by default (the library's choice is the baseline
    of the A/B); the line records what it ran under (`config.rccl_env`), so that one 8-GPU lease can compare
    `--rccl-proto LL`, `LL128` and `Simple` and read the difference off `rank_ms_per_step`."""
    if getattr(args, 'rccl_proto', None):
        env['NCCL_PROTO'] = args.rccl_proto
    if getattr(args, 'rccl_algo', None):
        env['NCCL_ALGO'] = args.rccl_algo
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # (the host driver only supports dmabuf IPC)
    return env


def rccl_env_record(env=None):
    env = os.environ if env is None else env
    return {k: env[k] for k in RCCL_ENV_KEYS if k in env}


def launch_command(args, port, argv=None):
    """The 8-rank (N-rank) command line of `python bench.py --gpus N`: one process per GPU under torch.distributed.run
    on the loopback, every rank re-running this script with the same arguments (each binds LOCAL_RANK -> its device in
    main())."""
    argv = sys.argv[1:] if argv is None else list(argv)
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
            '--master-addr', '127.0.0.1', '--master-port', str(port), str(Path(__file__).resolve())] + argv


def self_launch(args, run=None, n_visible=None):
    """`python bench.py --gpus N` (N > 1) without a launcher: start N fresh ranks of this script under
    torch.distributed.run and exit with their status. Runs BEFORE torch is imported: the parent neither
    maps nor initialises HIP (a process that has initialised it must never be replaced or forked into
    ranks), and it counts the GPUs from the KFD topology in sysfs. Refuses when fewer GPUs are visible than ranks
    asked for (unless PVS_BENCH_BACKEND names the dry-run backend). `run` / `n_visible`: injected by the CPU test."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = apply_rccl_choice(args, dict(os.environ))
    if n_visible is None:
        n_visible = visible_gpu_count()
    if n_visible is not None and n_visible < args.gpus and 'PVS_BENCH_BACKEND' not in env:
        raise SystemExit(f'--gpus {args.gpus} but only {n_visible} GPU(s) visible; set '
                         f'PVS_BENCH_BACKEND=gloo for a dry run of the multi-rank path with ranks sharing devices')
    cmd = launch_command(args, port)
    return (run or subprocess.run)(cmd, env=env).returncode


def cpu_baseline(cfg, seconds_budget=12.0):
    """The CPU oracle (oracle/egnn_oracle.py, a port of the reference's eager-PyTorch path) on
    the host cores: fwd + bwd + clip + Adam on batches of 1 and of 4 graphs of the same workload
    (SURVEY.md §8d protocol: eager autograd keeps ~13 GB of activations per layer at the full batch of 32,
    and the path is memory-bound, i.e. batch-size-insensitive); `value` is the better of the two."""
    from oracle import egnn_oracle as orc
    from pointvs_amd.synthetic import synthetic_graph
    from pointvs_amd.graph import Batch
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    torch.manual_seed(0)
    model = SartorrasEGNN(Path('/tmp/pvs_bench_cpu'), 2e-3, 1e-4, silent=True, **cfg['model'])
    sd0 = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    ocfg = dict(cfg['model'], _class='SartorrasEGNN')
    graphs = [synthetic_graph(1000 * cfg['cfg_id'] + k, **cfg['graph']) for k in range(4)]
    n_cpu = os.cpu_count() or 1

    def measure(n_graphs, budget, threads=None):
        g = Batch.from_data_list(graphs[:n_graphs])
        y_true = g.y.float()

        def one_step(state):
            _, _, grads = orc.forward_backward(state, ocfg, g.x, g.pos, g.edge_index, g.edge_attr,
                                               g.batch, y_true)
            new = orc.adam_step(state, grads, 2e-3, 1e-4)
            return {k: (new[k].numpy() if k in new else v) for k, v in state.items()}

        if threads is None:   # eager torch on one graph does not scale to a many-core host: pick the fastest count
            best_t = float('inf')
            for cand in sorted({min(n_cpu, t) for t in (8, 16, 32, 64)}):
                torch.set_num_threads(cand)
                one_step(sd0)
                t0 = time.perf_counter()
                one_step(sd0)
                dt = time.perf_counter() - t0
                if dt < best_t:
                    threads, best_t = cand, dt
        torch.set_num_threads(threads)
        sd, times, t_start = sd0, [], time.perf_counter()
        for it in range(12):
            t0 = time.perf_counter()
            sd = one_step(sd)
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > budget and it >= 2:
                break
        timed = times[1:] if len(times) > 1 else times
        return float(np.median(timed)), len(timed), threads, int(g.x.shape[0]), int(g.edge_index.shape[1])

    med1, n1, threads, nn, ne = measure(1, seconds_budget)
    med4, n4, _, _, _ = measure(4, seconds_budget, threads)
    rate1, rate4 = 1.0 / med1, 4.0 / med4
    return {'value': round(max(rate1, rate4), 4), 'unit': 'graphs/s', 'cores': threads, 'kind': 'port',
            'sample': f'CPU oracle, fwd+bwd+Adam, 1 warm-up each: {n1} timed steps on 1 graph (N={nn}, E={ne}): median '
                      f'{med1 * 1e3:.0f} ms/step = {rate1:.2f} graphs/s; {n4} timed steps on a 4-graph batch: median '
                      f'{med4 * 1e3:.0f} ms/step = {rate4:.2f} graphs/s; best of 8/16/32/64 torch threads on a '
                      f'{n_cpu}-CPU host, torch {torch.__version__} CPU'}


def cpu_baseline_screening(cfg, lig, rec, feats, seconds_budget=15.0):
    """The CPU oracle on the host cores for the virtual-screening shape: per pose, the oracle's
    generate_edges (cdist + the reference edge rule) and one forward of the model (torch.no_grad)."""
    from oracle import egnn_oracle as orc
    from oracle.generate_edges_oracle import generate_edges as oracle_edges
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    from pointvs_amd.synthetic import random_poses
    torch.manual_seed(0)
    model = SartorrasEGNN(Path('/tmp/pvs_bench_cpu'), 2e-3, 1e-4, silent=True, **cfg['model'])
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items() if v.is_floating_point()}
    ocfg = dict(cfg['model'], _class='SartorrasEGNN')
    r = cfg['graph']['edge_radius']
    bp = feats[:, -1].numpy()
    poses = random_poses(lig, 16, seed=99)
    n_cpu = os.cpu_count() or 1
    threads = min(n_cpu, 16)
    torch.set_num_threads(threads)
    batch = torch.zeros(lig.shape[0] + rec.shape[0], dtype=torch.long)

    def one_pose(p):
        pos = torch.cat([p, rec], 0)
        _, (rows, cols), attrs = oracle_edges(pos.numpy(), bp, r, r, prune=False)
        ei = torch.from_numpy(np.vstack([rows, cols])).long()
        ea = torch.nn.functional.one_hot(torch.from_numpy(attrs).long(), 3)
        with torch.no_grad():
            return orc.model_forward(sd, ocfg, feats, pos, ei, ea, batch, n_graphs=1), ei.shape[1]

    one_pose(poses[0])
    times, t_start, n_edges = [], time.perf_counter(), 0
    for p in poses[1:]:
        t0 = time.perf_counter()
        _, n_edges = one_pose(p)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > seconds_budget and len(times) >= 3:
            break
    med = float(np.median(times))
    return {'value': round(1.0 / med, 4), 'unit': 'graphs/s', 'cores': threads, 'kind': 'port',
            'sample': f'{len(times)} timed poses (1 warm-up) of the CPU oracle: radius graph (cdist + reference edge '
                      f'rule, E={n_edges}) + forward of the same model on N={batch.numel()} atoms, median '
                      f'{med * 1e3:.0f} ms/pose, {threads} torch threads on a {n_cpu}-CPU host, torch '
                      f'{torch.__version__} CPU'}


def screening_bench(args, rank, world, dev):
    """BASELINE config 5: poses/s of the forward pass (torch.no_grad) with the radius graph of every
    pose built on the GPU from its coordinates, the whole step replayed from a hipGraph, scores
    streamed to a predictions file by the writer thread. One step = one batch of `--batch` poses;
    `--sweep P` screens P poses per rank instead of `--steps` batches (BASELINE: 100k over 8 GPUs =
    12.5k per rank). Every rank screens its own poses (no collective: pose shards are independent)."""
    from pointvs_amd import _lib
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    from pointvs_amd.screening import ScreeningSweep
    from pointvs_amd.synthetic import CONFIGS, random_poses, screening_set
    cfg = CONFIGS['cfg2']
    lib = _lib.lib()
    lig, rec, feats = screening_set()
    n_lig = lig.shape[0]
    torch.manual_seed(0)
    model = SartorrasEGNN(Path('/tmp/pvs_bench'), 2e-3, 1e-4, silent=True, **cfg['model']).eval()
    if args.sweep:
        args.steps = -(-args.sweep // args.batch)
    n_warm = max(args.warmup, 1)
    poses = random_poses(lig, (n_warm + args.steps) * args.batch, seed=7 + rank, device=dev)
    warm, timed = poses[:n_warm * args.batch], poses[n_warm * args.batch:]
    captured = bool(args.graph)
    sweep = ScreeningSweep(model, rec.to(dev), feats[n_lig:].to(dev), cfg['graph']['edge_radius'], args.batch,
                           capture=captured)
    lig_feats = feats[:n_lig]
    out_dir = Path(os.environ.get('PVS_BENCH_OUT', '/tmp/pvs_bench'))
    pred_file = out_dir / f'screen_predictions_rank{rank}.txt'

    sweep.run([('warm', lig_feats, warm)])           # builds the bucket, captures the step
    if getattr(args, 'distributed', world > 1):
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    scores = sweep.run([('lig0', lig_feats, timed)], predictions_file=pred_file)['lig0']
    torch.cuda.synchronize(dev)
    if getattr(args, 'distributed', world > 1):
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if getattr(args, 'distributed', world > 1):
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_lines = sum(1 for _ in open(pred_file))
    assert n_lines == timed.shape[0], (n_lines, timed.shape[0])

    # kernel timings of the same step, eager (HIP events cannot be read out of a replayed graph)
    screen = sweep.buckets[n_lig]
    lib.pvs_profile_reset()
    lib.pvs_profile_enable(1)
    prof_steps = 5
    for k in range(prof_steps):
        screen(timed[k * args.batch:(k + 1) * args.batch].contiguous())
    torch.cuda.synchronize(dev)
    lib.pvs_profile_enable(0)

    def kernel_ms(name):
        tot, cnt = C.c_double(0.0), C.c_int64(0)
        rc = lib.pvs_profile_read(name.encode(), C.byref(tot), C.byref(cnt))
        return (tot.value, cnt.value) if rc == 0 else (0.0, 0)

    if rank == 0:
        h, layers = cfg['model']['k'], cfg['model']['num_layers']
        n_nodes = args.batch * (n_lig + rec.shape[0])
        n_edges = int(screen._fast['rowptr'][n_nodes].item()) if screen._fast else screen.batcher.batch.prepared.n_edges
        n_edges_lig = int(screen._fast['rowptr_l'][n_nodes].item()) if screen._fast else 0
        fwd_ms, fwd_n = kernel_ms('edge_fwd')
        part_ms, part_n = kernel_ms('edge_fwd_partial')
        prep_ms, _ = kernel_ms('graph_prepare')
        dom_avg_ms = fwd_ms / max(fwd_n, 1)
        dom_bytes = algorithmic_bytes_edge_fwd(n_nodes, n_edges, h)
        achieved = dom_bytes / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else 0.0
        dom_tflops = (4.0 * h * h + 2 * h) * n_edges / (dom_avg_ms * 1e-3) / 1e12 if dom_avg_ms > 0 else 0.0
        traffic, traffic_src = measured_traffic('cfg5', 'k_edge_fwd_mfma')
        ms_step = elapsed / args.steps * 1e3
        out = {
            'metric': 'ligand poses/sec, forward only (virtual-screening sweep), 3-layer EGNN ch=32, one '
                      'receptor of 1970 atoms, r=10A, graphs built on the GPU',
            'value': round(world * timed.shape[0] / elapsed, 2), 'unit': 'graphs/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': n_warm, 'ms_per_step': round(ms_step, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'cfg5: {timed.shape[0]} poses/GPU streamed in batches of {args.batch}, 30-atom '
                                   f'ligand + 1970-atom receptor, N={n_nodes} nodes E={n_edges} edges per batch '
                                   f'({n_edges_lig} touch the ligand), radius graph from coordinates every step, '
                                   f'first-layer receptor-receptor sums reused: {screen.reuse}, hipGraph replay: '
                                   f'{bool(screen._captured)}, predictions streamed to a file ({n_lines} lines)',
                       'graphs_per_gpu': args.batch, 'global_batch': world * args.batch, 'parallelism': f'dp{world}',
                       'mean_score': round(float(scores.mean()), 6)},
            'roofline': {
                'bound': 'hbm', 'kernel': f'k_edge_fwd_mfma (H={h} edge forward over the full pose-batch graph, '
                                          f'{layers - 1} launches per step; the first layer runs over the ligand edges only)',
                'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic, 'traffic_source': traffic_src,
                'algorithmic_bytes_per_launch': dom_bytes, 'avg_launch_ms': round(dom_avg_ms, 4), 'launches': fwd_n,
                'kernel_exec_fp32': {'achieved': round(dom_tflops, 2), 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                     'frac': round(dom_tflops / FP32_PEAK_TFLOPS, 5)},
                'kernel_ms_per_step': {'edge_fwd_full_layers': round(fwd_ms / prof_steps, 3),
                                       'edge_fwd_first_layer_ligand_edges': round(part_ms / prof_steps, 3),
                                       'graph_build': round(prep_ms / prof_steps, 3)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline_screening(cfg, lig, rec, feats)
        return out
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='cfg2', choices=['cfg2', 'cfg3', 'cfg5', 'real4A'],
                    help='cfg5: virtual-screening sweep (BASELINE config 5): forward only, random poses '
                         'of one ligand against one receptor, graphs built on the GPU per batch; real4A: NOT a BASELINE '
                         'configuration - the reference\'s default CLI shape (6 layers, 32 channels, r = 4 A, ~500 atoms '
                         'per graph: the launch-bound regime; compare --graph 0 and --graph 1)')
    ap.add_argument('--batch', type=int, default=None,
                    help='graphs per GPU (default 32; cfg5: poses per replayed step, default 128 - the sweep streams '
                         'one ligand\'s poses in fixed-size batches of its choosing: 24.5k / 26.2k / 27.7k / 26.8k poses/s '
                         'at 32 / 64 / 128 / 256, profiles/r03_cfg5_batch_sizes.txt)')
    ap.add_argument('--sweep', type=int, default=0,
                    help='cfg5: screen this many poses per rank (sets --steps = ceil(sweep / batch)); BASELINE '
                         'config 5 is 100k poses over 8 GPUs')
    ap.add_argument('--global-batch', type=int, default=0,
                    help='strong scaling (BASELINE config 4): fixed global batch, --batch becomes global/gpus')
    ap.add_argument('--skip-dead-coords', action='store_true',
                    help='let the model skip the last layer\'s coordinate update, whose result nothing reads '
                         '(the library default); the bench evaluates it by default, like the reference does')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true',
                    help='default cfg2 run only: do not also time BASELINE configs 3 and 5 behind the headline (the '
                         '"secondary" object of the line); implied by --no-cpu-baseline, the flag of every profiling and '
                         'A/B invocation under tools/ (a kernel trace of ONE configuration)')
    ap.add_argument('--model-flags', default='',
                    help='NOT a BASELINE configuration: extra model keywords on top of --config, e.g. '
                         '"edge_residual=True,tanh=True" (profiles of kernel instantiations no BASELINE config uses); '
                         'the workload string names them')
    ap.add_argument('--infer', action='store_true',
                    help='forward-only (torch.no_grad) throughput: the virtual-screening shape of '
                         'BASELINE config 5; not the headline metric')
    ap.add_argument('--build-graph', action='store_true',
                    help='build the radius graph on the GPU from the coordinates every step '
                         '(pvs_radius_graph_*, SURVEY §8f row 1) instead of parsing the resident int64 '
                         'COO + one-hot; not the headline configuration (one host sync per step)')
    ap.add_argument('--host-inputs', choices=['pinned', 'pageable'], default=None,
                    help='hand every step a HOST batch (as the reference\'s DataLoader does) and count the '
                         'host-to-device copy in the step: the PCIe-inclusive rate noted in DESIGN.md, never '
                         'the headline value (inputs resident in HBM)')
    ap.add_argument('--force-dist', action='store_true',
                    help='with --gpus 1: create a process group of ONE rank and run every collective of the '
                         'multi-rank path anyway (hooks, bucketed all-reduce, barriers, max-over-ranks): an '
                         'RCCL smoke on a one-GPU box; not a measured configuration')
    ap.add_argument('--rccl-proto', choices=['LL', 'LL128', 'Simple'], default=None,
                    help='multi-rank runs: NCCL_PROTO for the gradient all-reduce (92 KB at cfg2 / cfg4, 1.42 MB at cfg3: '
                         'latency-bound), set before the process group is created; default: RCCL\'s own choice. The line '
                         'records the RCCL environment it ran under (config.rccl_env)')
    ap.add_argument('--rccl-algo', choices=['Ring', 'Tree'], default=None, help='NCCL_ALGO, likewise')
    ap.add_argument('--graph', type=int, default=None,
                    help='1: capture the whole step in a hipGraph and time replays (default: 1 for cfg5, the '
                         'configuration BASELINE names; 0 for the training configurations)')
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 128 if args.config == 'cfg5' else 32
    if args.graph is None:
        args.graph = int(os.environ.get('PVS_BENCH_GRAPH', '1' if args.config == 'cfg5' else '0'))
    strong = args.global_batch > 0
    if not args.skip_dead_coords:      # same per-step work as the reference: every layer updates x
        os.environ['PVS_EGNN_KEEP_DEAD_COORDS'] = '1'
    if strong:
        if args.global_batch % args.gpus:
            raise SystemExit(f'--global-batch {args.global_batch} is not divisible by --gpus {args.gpus}')
        args.batch = args.global_batch // args.gpus

    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(self_launch(args))
    apply_rccl_choice(args, os.environ)        # (a rank started by another launcher: still before init_process_group)
    _import_torch()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (there is no CPU path in the product)')
    # PVS_BENCH_BACKEND=gloo (+ ranks sharing the visible GPUs): a dry run of the multi-rank code path
    # on a box with fewer GPUs than ranks; the measured configuration is always one rank per GPU over RCCL
    backend = os.environ.get('PVS_BENCH_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    distributed = world > 1 or args.force_dist
    args.distributed = distributed
    if distributed:
        if world == 1:      # --force-dist: a group of one, rendezvous on the loopback
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if 'MASTER_PORT' not in os.environ:      # a free port, as self_launch() picks one (two smoke runs on one box)
                import socket
                with socket.socket() as sock:
                    sock.bind(('127.0.0.1', 0))
                    os.environ['MASTER_PORT'] = str(sock.getsockname()[1])
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    if args.config == 'cfg5':
        out = screening_bench(args, rank, world, dev)
    else:
        out = training_bench(args, rank, world, dev)
        plain = not (args.infer or args.model_flags or args.build_graph or args.host_inputs or strong or args.force_dist
                     or args.graph or args.skip_dead_coords or args.batch != 32)
        if args.config == 'cfg2' and world == 1 and plain and not (args.no_secondary or args.no_cpu_baseline):
            sec = secondary_configs(args, rank, world, dev)
            if out is not None:
                out['secondary'] = sec
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def secondary_configs(args, rank, world, dev):
    """BASELINE configs 3 and 5 in the SAME process as the headline line, behind its timed region (VERDICT r04 item 2:
    the driver runs only `bench.py --gpus 1 --steps 20 --warmup 5`, so these two were builder-run numbers only):
    cfg3 = 10 timed steps of the 12-layer / 64-channel / edge + node attention model, cfg5 = a 12,800-pose sweep (the
    per-GPU share of BASELINE's 100k poses over 8 GPUs) with the step replayed from a hipGraph. Same code paths as
    `--config cfg3` / `--config cfg5 --sweep 12800`; the full records of those invocations are in profiles/."""
    import copy
    import gc
    sec = {}
    # (round 6: also the reference's own default shape - NOT a BASELINE configuration, the launch-bound regime - eager)
    for name, changes in (('cfg3', dict(config='cfg3', steps=10, warmup=3, batch=32, graph=0)),
                          ('cfg5', dict(config='cfg5', sweep=12800, warmup=1, batch=128, graph=1)),
                          ('real4A', dict(config='real4A', steps=200, warmup=30, batch=32, graph=0))):
        gc.collect()
        torch.cuda.empty_cache()
        a = copy.copy(args)
        for k, v in changes.items():
            setattr(a, k, v)
        a.no_cpu_baseline = True
        rec = screening_bench(a, rank, world, dev) if name == 'cfg5' else training_bench(a, rank, world, dev)
        if rec is None:
            continue
        roof = rec['roofline']
        sec[name] = {'metric': rec['metric'], 'value': rec['value'], 'unit': rec['unit'], 'steps': rec['steps'],
                     'warmup': rec['warmup'], 'ms_per_step': rec['ms_per_step'], 'workload': rec['config']['workload'],
                     'layer_calls': rec['config'].get('layer_calls'),
                     'roofline': {k: roof.get(k) for k in ('kernel', 'bound', 'frac', 'achieved', 'avg_launch_ms',
                                                           'avg_launch_ms_full_work', 'frac_full_work', 'launches',
                                                           'algorithmic_bytes_per_launch', 'kernel_ms_per_step')}}
    return sec


def training_bench(args, rank, world, dev):
    """One training (or --infer) configuration: returns the record on rank 0, None elsewhere."""
    from pointvs_amd import _lib, functional as PF, graph as pgraph
    from pointvs_amd.distributed import OverlappedGradAllReducer
    from pointvs_amd.egnn_satorras import SartorrasEGNN
    from pointvs_amd.synthetic import CONFIGS, synthetic_batch
    distributed = args.distributed
    strong = args.global_batch > 0
    cfg = CONFIGS[args.config]
    if args.model_flags:
        import ast
        extra = {k.strip(): ast.literal_eval(v.strip()) for k, v in (kv.split('=', 1) for kv in args.model_flags.split(','))}
        cfg = dict(cfg, model=dict(cfg['model'], **extra))
    lib = _lib.lib()
    cache_was = pgraph.CACHE_ENABLED
    pgraph.CACHE_ENABLED = False        # every step prepares its batch, as a fresh batch would

    # ---- inputs: this rank's graphs, built on the host, then resident in HBM ----
    batch = synthetic_batch(cfg['cfg_id'], args.batch, first_graph=rank * args.batch, **cfg['graph'])
    n_nodes, n_edges = int(batch.x.shape[0]), int(batch.edge_index.shape[1])
    host_batch = None
    if args.host_inputs:
        import copy
        host_batch = copy.copy(batch)       # Data.to() moves in place: keep a separate attribute bag on the host
        if args.build_graph:      # only coordinates and features cross PCIe; the edges are built on the GPU
            for name in ('edge_index', 'edge_attr'):
                delattr(host_batch, name)
        if args.host_inputs == 'pinned':
            for name in host_batch.keys():
                v = getattr(host_batch, name)
                if torch.is_tensor(v):
                    setattr(host_batch, name, v.pin_memory())
    batch = batch.to(dev)
    y_true = batch.y.float()

    torch.manual_seed(0)
    model = SartorrasEGNN(Path('/tmp/pvs_bench'), 2e-3, 1e-4, silent=True, **cfg['model']).train()
    params = list(model.parameters())
    use_graph = bool(args.graph) and not distributed and not args.build_graph and not args.host_inputs
    if use_graph:   # same Adam, step counter kept on the device so the step can be captured
        from pointvs_amd.optim import FusedClipAdam
        cls = torch.optim.Adam if os.environ.get('PVS_BENCH_TORCH_ADAM') else FusedClipAdam
        model.optimiser = cls(params, lr=2e-3, weight_decay=1e-4, capturable=True)
    # exchange of the late layers' gradients starts from backward hooks, the rest after the backward
    elif os.environ.get('PVS_BENCH_TORCH_ADAM'):   # A/B: torch's multi-tensor Adam + clip_grad_value_
        model.optimiser = torch.optim.Adam(params, lr=2e-3, weight_decay=1e-4)
    reducer = OverlappedGradAllReducer(params, exchange_when_alone=True) if distributed else None

    # measured on MI355X: no gain (8.9 ms with and without; the sorts contend with the edge
    # kernels), so off by default
    prefetch = int(os.environ.get('PVS_BENCH_PREFETCH', '0')) and not use_graph

    def infer_step():
        with torch.no_grad():
            return model(batch).reshape(-1).sum()

    if args.build_graph:
        from pointvs_amd.radius_graph import attach_radius_graph

    def step():
        nonlocal batch
        if host_batch is not None:
            batch = copy.copy(host_batch).to(dev, non_blocking=True)
        if args.build_graph:
            attach_radius_graph(batch, cfg['graph']['edge_radius'])
        if args.infer:
            return infer_step()
        y_pred = model(batch).reshape(-1)
        if prefetch:   # the next batch's CSR/CSC build (here: the same tensors) runs on a side
            # stream under this batch's backward, as a data loader's look-ahead would arrange it
            pgraph.prefetch_graph(batch.edge_index, batch.edge_attr, n_nodes, layout=pgraph.runs_layout(batch))
        loss = model.get_loss(y_true, y_pred)
        model.optimiser.zero_grad()
        # (as the harness's backprop() calls it: loss.backward() with the root gradient 1 handed over instead of filled)
        loss.backward(gradient=PF.unit_gradient(loss.device))
        if reducer is not None:
            reducer()
        if hasattr(model.optimiser, '_fusable'):      # pointvs_amd.optim.FusedClipAdam: clip + Adam in one launch
            model.optimiser.step(clip_value=1.0)
        else:
            torch.nn.utils.clip_grad_value_(params, 1.0)
            model.optimiser.step()
        return loss

    # Graph mode keeps every step (warm-up, capture, replays, the profiled eager steps) on ONE
    # non-default stream: autograd's AccumulateGrad nodes remember the stream of their first
    # backward, and a capture on a different stream than earlier eager steps faults in
    # hipStreamEndCapture (ROCm 7.2 / torch 2.10).
    work_stream = torch.cuda.Stream(dev) if use_graph else torch.cuda.current_stream(dev)
    with torch.cuda.stream(work_stream):
        for _ in range(max(args.warmup, 2 if use_graph else 0)):
            step()
        eager_step = step
        if use_graph:
            torch.cuda.synchronize(dev)
            hip_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(hip_graph, stream=work_stream):
                static_loss = eager_step()

            def step():   # noqa: F811  (one graph launch = one full training step)
                hip_graph.replay()
                return static_loss
            step()
        # HIP events around the DOMINANT kernel only inside the timed region (roofline.avg_launch_ms is measured live, as
        # the contract asks); an event pair costs the stream a ~6 us bubble per launch (tools/step_timeline.py), so the
        # other groups of the per-step breakdown are timed in three more steps behind the timed region
        prof_ids = {'edge_fwd': 0, 'edge_bwd': 1, 'col_gather': 2, 'graph_prepare': 3}
        dom_group = 'edge_fwd' if args.infer else 'edge_bwd'
        lib.pvs_profile_reset()
        lib.pvs_profile_enable(0 if use_graph else 1 << (prof_ids[dom_group] + 1))
        # (as train_model does, point_neural_network_base.long_lived_heap_frozen: Python's first full collection walks the
        # ~170,000 objects `import torch` leaves behind - 60-170 ms, once, wherever the allocation count puts it: in a
        # 0.1 s timed region that is luck, not throughput. PVS_GC_FREEZE=0 leaves the collector alone.)
        import gc
        freeze = os.environ.get('PVS_GC_FREEZE') != '0'
        if freeze:
            gc.freeze()
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        gc_log, step_host = [], []
        tracing = bool(os.environ.get('PVS_BENCH_TRACE_STEPS'))      # (diagnostic: host time of every timed step and every
        if tracing:                                                  # collector pause of a millisecond or more, to stderr)
            gc.callbacks.append(lambda phase, info: gc_log.append((phase, info.get('generation'), time.perf_counter())))
        t0 = time.perf_counter()
        for _ in range(args.steps):
            if tracing:
                ts = time.perf_counter()
                loss = step()
                step_host.append((time.perf_counter() - ts) * 1e3)
            else:
                loss = step()
        t_host = time.perf_counter()
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if tracing:
            gc.callbacks.pop()
            print(f'host ms per timed step: median {sorted(step_host)[len(step_host) // 2]:.3f}, the slowest '
                  f'{sorted(((round(v, 2), k) for k, v in enumerate(step_host)), reverse=True)[:6]}, host loop '
                  f'{(t_host - t0) * 1e3:.1f} ms of {elapsed * 1e3:.1f} ms', file=sys.stderr)
            starts = {}
            for phase, gen, t in gc_log:
                if phase == 'start':
                    starts[gen] = t
                elif gen in starts and (t - starts[gen]) > 1e-3:
                    print(f'  gc generation {gen}: {(t - starts[gen]) * 1e3:.1f} ms at +{(starts[gen] - t0) * 1e3:.1f} ms',
                          file=sys.stderr)
        if freeze:
            gc.unfreeze()
        lib.pvs_profile_enable(0)
        dom_live = None if use_graph else _kernel_ms(lib, dom_group)      # (total ms, launches) of the timed region
        dom_each = [] if use_graph else _kernel_each(lib, dom_group)
        # the per-step breakdown (every group), eager, behind the timed region
        lib.pvs_profile_reset()
        lib.pvs_profile_enable(1)
        prof_steps = 3
        for _ in range(prof_steps):
            eager_step()
        torch.cuda.synchronize(dev)
        lib.pvs_profile_enable(0)
    rank_ms = [elapsed / args.steps * 1e3]
    if distributed:     # the contract's value uses the MAX over ranks; min / max per rank make a straggler visible
        every = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(every, torch.tensor([elapsed], dtype=torch.float64, device=dev))
        rank_ms = [float(v.item()) / args.steps * 1e3 for v in every]
        elapsed = max(float(v.item()) for v in every)
    final_loss = float(loss.item())
    pgraph.CACHE_ENABLED = cache_was

    def kernel_ms(name):
        return _kernel_ms(lib, name)

    if rank == 0:
        h = cfg['model']['k']
        layers = cfg['model']['num_layers']
        eatt, natt = cfg['model']['edge_attention'], cfg['model']['node_attention']
        training = not args.infer
        ms_step = elapsed / args.steps * 1e3
        graphs_per_s = world * args.batch * args.steps / elapsed
        bwd_ms, bwd_n = kernel_ms('edge_bwd')
        fwd_ms, fwd_n = kernel_ms('edge_fwd')
        col_ms, col_n = kernel_ms('col_gather')
        prep_ms, prep_n = kernel_ms('graph_prepare')
        if training:     # dominant kernel: the edge backward (one launch per layer)
            dom_ms, dom_n = dom_live if dom_live and dom_live[1] > 0 else (bwd_ms, bwd_n)
            dom_bytes = algorithmic_bytes_edge_bwd(n_nodes, n_edges, h)
            dom_flops = (12.0 * h * h + 4 * h) * n_edges     # 2 recompute + 2 dgrad + 2 wgrad products
            fp32_family = os.environ.get('PVS_EGNN_BF16X3') == '0'
            dom_symbol = (('k_edge_bwd_mfma' if fp32_family else 'k_edge_bwd_f16') if h == 32
                          else ('k_edge_bwd_team' if fp32_family else 'k_edge_bwd_h64'))
            dom_name = (f'{dom_symbol} (H={h} edge backward, one launch per layer)')
            step_bytes = layers * algorithmic_bytes_per_layer(n_nodes, n_edges, h)
        else:            # forward only: the edge forward
            dom_ms, dom_n = dom_live if dom_live and dom_live[1] > 0 else (fwd_ms, fwd_n)
            dom_bytes = algorithmic_bytes_edge_fwd(n_nodes, n_edges, h)
            dom_flops = (4.0 * h * h + 2 * h) * n_edges
            dom_symbol = 'k_edge_fwd_mfma'
            dom_name = f'k_edge_fwd_mfma (H={h} edge forward, one launch per layer)'
            step_bytes = layers * dom_bytes
        dom_avg_ms = dom_ms / max(dom_n, 1)
        achieved = dom_bytes / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else 0.0
        dom_tflops = dom_flops / (dom_avg_ms * 1e-3) / 1e12 if dom_avg_ms > 0 else 0.0
        # The launches of one step do not do the same work: the backward runs the layers last to first, and nothing
        # differentiates the last layer's coordinate update (its x is dead: SURVEY Q3, the reference's autograd skips
        # it too), so the FIRST backward launch of every step has no coordinate branch. `avg_launch_ms` is over all
        # launches (the contract's definition); `avg_launch_ms_full_work` leaves those lighter launches out.
        full = [ms for k, ms in enumerate(dom_each) if k % layers != 0] if (training and layers > 1) else []
        full_avg_ms = sum(full) / len(full) if full else None
        ref_flops = algorithmic_flops_per_step(n_nodes, n_edges, h, 3, layers, eatt, natt, training)
        exec_flops = executed_flops_per_step(n_nodes, n_edges, h, layers, eatt, natt, training)
        traffic, traffic_src = measured_traffic(args.config, dom_symbol)
        what = 'forward only (inference)' if args.infer else 'fwd+bwd (+Adam step)'
        shape = {'cfg2': '3-layer EGNN ch=32, ~2k nodes r=10A',
                 'cfg3': '12-layer EGNN ch=64 edge+node attention, ~2k nodes r=6A',
                 'real4A': 'NOT A BASELINE CONFIGURATION: 6-layer EGNN ch=32, ~500 nodes r=4A (reference CLI defaults)'}[args.config]
        out = {
            'metric': f'protein-ligand graphs/sec {what}, {shape}',
            'value': round(graphs_per_s, 2), 'unit': 'graphs/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_step, 3),
            'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': (f'NOT A BASELINE CONFIGURATION ({args.config} + --model-flags {args.model_flags}): '
                                    if args.model_flags else '') +
                                   f'{args.config}: {layers}-layer EGNN, channels={h}, '
                                   f'edge_radius={cfg["graph"]["edge_radius"]}A, '
                                   f'{args.batch} graphs/GPU x {cfg["graph"]["n_nodes"]} atoms, '
                                   f'N={n_nodes} nodes E={n_edges} edges per rank, '
                                   f'{flag_summary(cfg["model"])}, '
                                   + ('torch.no_grad forward' if args.infer else
                                      'Adam lr 2e-3 wd 1e-4 clip 1.0') + ', random init',
                       'graphs_per_gpu': args.batch, 'global_batch': world * args.batch,
                       'parallelism': f'dp{world}', 'final_loss': round(final_loss, 6),
                       'arithmetic': ARITHMETIC,
                       # self-diagnosis of a multi-rank run: what the process group really is, and
                       # every rank's own step time beside the max the value is computed from
                       'rccl_world': dist.get_world_size() if distributed else 1,
                       'dist_backend': dist.get_backend() if distributed else None,
                       'rccl_version': ('.'.join(str(v) for v in torch.cuda.nccl.version())
                                        if distributed and dist.get_backend() == 'nccl' else None),
                       'rccl_env': rccl_env_record() if distributed else None,
                       'rank_ms_per_step': {'min': round(min(rank_ms), 3), 'max': round(max(rank_ms), 3),
                                            'all': [round(v, 3) for v in rank_ms]},
                       'scaling_note': scaling_note(args, world, strong),
                       'last_layer_coord_update': 'skipped (dead)' if args.skip_dead_coords else 'evaluated',
                       'launch': 'hipGraph replay of the whole step' if use_graph else 'eager',
                       'host_gc': ('heap frozen (gc.freeze) for the timed region, as train_model does' if freeze else
                                   'Python collector left alone (PVS_GC_FREEZE=0)'),
                       'layer_calls': ('one call each way for the whole layer stack (pvs_egnn_stack_fwd / _bwd)'
                                       if model.__dict__.get('_stack_cache') is not None else
                                       'one call each way per layer (pvs_egnn_layer_fwd / _bwd)'),
                       'inputs': (f'host ({args.host_inputs}) batch copied to the device inside every step'
                                  if args.host_inputs else 'resident in HBM'),
                       'graph_prepare': ('radius graph built on the GPU from the coordinates' if args.build_graph else
                                         'int64 COO + one-hot parsed every step; generate_edges-ordered batch merged by '
                                         'counting (verified on the device)'
                                         if pgraph.runs_layout(batch) is not None else
                                         'int64 COO + one-hot parsed every step (two radix sorts)')},
            'roofline': {
                'bound': 'hbm', 'kernel': dom_name,
                'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic, 'traffic_source': traffic_src,
                **measured_limiter(args.config, dom_symbol),
                'algorithmic_bytes_per_launch': dom_bytes,
                'avg_launch_ms': round(dom_avg_ms, 4), 'launches': dom_n,
                'avg_launch_ms_full_work': None if full_avg_ms is None else round(full_avg_ms, 4),
                'frac_full_work': None if not full_avg_ms else round(dom_bytes / (full_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                # the kernel is ALU-bound, not HBM-bound: its EXECUTED products against the
                # fp32 matrix peak, beside the HBM fraction north_star asks for. The products run as three fp16
                # terms (H = 32) or six bf16 terms (H = 64 backward) on the 16-bit matrix pipes, so this is an
                # effective fp32 rate, not a utilisation: the PMC busy figures in profiles/ are the utilisation
                'kernel_exec_fp32': {'achieved': round(dom_tflops, 2), 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                     'frac': round(dom_tflops / FP32_PEAK_TFLOPS, 5)},
                'step_hbm_frac': round(step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                # executed FLOPs of the step (the P/Q split removes 4H^2 per edge per pass of the
                # reference formulation) and, for comparison, the reference formulation's count
                'step_exec_fp32_frac': round(exec_flops / (ms_step * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 5),
                'step_ref_formulation_fp32_frac': round(ref_flops / (ms_step * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 5),
                'kernel_ms_per_step': {
                    'edge_fwd': round(fwd_ms / prof_steps, 3),
                    'edge_bwd': round(bwd_ms / prof_steps, 3),
                    'col_gather': round(col_ms / prof_steps, 3),
                    'graph_prepare': round(prep_ms / prof_steps, 3)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg)
        return out
    return None


if __name__ == '__main__':
    main()
