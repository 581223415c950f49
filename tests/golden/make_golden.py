#!/usr/bin/env python3
"""Generate golden input/output vectors by importing the real reference (build container only).

Run from the repo root, in the container that has `/root/reference`:

    python tests/golden/make_golden.py          # all cases
    python tests/golden/make_golden.py --add    # only cases without a committed .npz

It imports `/root/reference/point_vs` under the import-only stand-ins in `tests/golden/_refstubs`
(see the README there), drives the reference's own classes
(`SartorrasEGNN`, `MultitaskSatorrasEGNN`, its data loader and `generate_edges`) on fixed seeds and
writes one compressed `.npz` per case into `tests/golden/`. The `.npz` files are data only:
inputs, the random-init `state_dict`, and what the reference computed from them (per-layer node
features / coordinates, final edge messages, attention side-attributes, logits, every parameter
gradient of the loss, the parameters after one `backprop()` step).

Nothing on the GPU box ever runs this script or reads `/root/reference`.
"""
import json
import os
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
REF = Path('/root/reference')
sys.path.insert(0, str(HERE / '_refstubs'))
sys.path.insert(0, str(REF))

import numpy as np  # noqa: E402

if not hasattr(np, 'product'):  # NumPy 2 removed the alias the reference still calls
    np.product = np.prod
if not hasattr(np, 'alltrue'):
    np.alltrue = np.all

import torch  # noqa: E402

os.chdir(REF)  # the reference's test fixtures use relative paths

from point_vs.models.geometric.egnn_satorras import SartorrasEGNN  # noqa: E402
from point_vs.models.geometric.egnn_multitask import MultitaskSatorrasEGNN  # noqa: E402
from point_vs.models.geometric.egnn_satorras import EGNNLayer  # noqa: E402
from point_vs.preprocessing.data_loaders import get_data_loader, PygPointCloudDataset  # noqa: E402
from point_vs.preprocessing.preprocessing import generate_edges, uniform_random_rotation  # noqa: E402
from torch_geometric.data import Batch, Data  # noqa: E402  (stub)
import pandas as pd  # noqa: E402

torch.set_num_threads(1)  # summation order of the CPU kernels must not depend on the host


def reference_test_graphs():
    """G1/G2/G3: the reference's own unit-test graphs (test/setup_and_params.py:15-58)."""
    common = dict(
        dataset_class=PygPointCloudDataset, compact=True, radius=4, use_atomic_numbers=False,
        rot=False, augmented_actives=0, min_aug_angle=0, polar_hydrogens=False, receptors=None,
        mode='val', types_fname=Path('test/resources/test.types'), fname_suffix='.parquet',
        edge_radius=4, estimate_bonds=True)
    one = next(iter(get_data_loader(Path('test/resources'), batch_size=1, **common)))
    two = next(iter(get_data_loader(Path('test/resources'), batch_size=2, **common)))
    np.random.seed(2)
    rot_pos = torch.from_numpy(uniform_random_rotation(one.pos.numpy().copy())).float()
    rotated = Batch(x=one.x, edge_index=one.edge_index, edge_attr=one.edge_attr, pos=rot_pos,
                    batch=one.batch.clone(), y=one.y, lig_fname=one.lig_fname,
                    rec_fname=one.rec_fname)
    return one, two, rotated


def synthetic_ball_graph(n_nodes, n_lig, edge_radius, seed, density=0.05):
    """G4: uniform ball pushed through the reference's generate_edges (duplicate inter edges)."""
    rng = np.random.default_rng(seed)
    big_r = (3.0 * n_nodes / (4.0 * np.pi * density)) ** (1.0 / 3.0)
    pts = rng.normal(size=(n_nodes, 3))
    pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    pts *= big_r * rng.random(n_nodes)[:, None] ** (1.0 / 3.0)
    order = np.argsort(np.linalg.norm(pts, axis=1))
    pts = pts[order].astype(np.float32)
    bp = np.ones(n_nodes, dtype=np.int64)
    bp[:n_lig] = 0
    struct = pd.DataFrame({'x': pts[:, 0], 'y': pts[:, 1], 'z': pts[:, 2], 'bp': bp,
                           'types': rng.integers(0, 11, n_nodes)})
    struct, edge_idx, edge_attr = generate_edges(
        struct, inter_radius=edge_radius, intra_radius=edge_radius, prune=False)
    feats = np.zeros((n_nodes, 12), dtype=np.float32)
    feats[np.arange(n_nodes), struct['types'].to_numpy()] = 1.0
    feats[:, 11] = bp
    return Data(
        x=torch.from_numpy(feats),
        edge_index=torch.from_numpy(np.vstack(edge_idx)).long(),
        edge_attr=torch.nn.functional.one_hot(torch.from_numpy(edge_attr).long(), 3),
        pos=torch.from_numpy(pts), y=torch.tensor(1), lig_fname='lig', rec_fname='rec',
        dE=None, rmsd=None)


def clone_graph(g):
    return Batch(x=g.x.clone(), edge_index=g.edge_index.clone(), edge_attr=g.edge_attr.clone(),
                 pos=g.pos.clone(), batch=g.batch.clone(), y=g.y.clone(),
                 lig_fname=g.lig_fname, rec_fname=g.rec_fname)


def config1_graphs():
    """BASELINE config 1 / the README working example (README.md:56-65): the first batches the
    reference's own loader yields on data/small_chembl_test (pose) and
    data/multi_classification_sample (affinity) with the CLI defaults of point_vs.py:108-121
    (radius 10, edge_radius 4, no --compact => 22 input features, no --estimate_bonds)."""
    common = dict(
        dataset_class=PygPointCloudDataset, batch_size=3, compact=False, radius=10,
        use_atomic_numbers=False, rot=False, polar_hydrogens=False, fname_suffix='parquet',
        edge_radius=4.0, estimate_bonds=False, prune=False, extended_atom_types=False,
        include_strain_info=False, mode='val')
    pose = next(iter(get_data_loader(
        Path('data/small_chembl_test'), types_fname=Path('data/small_chembl_test.types'),
        model_task='classification', **common)))
    affinity = next(iter(get_data_loader(
        Path('data/multi_classification_sample'),
        types_fname=Path('data/multi_classification_sample.types'), model_task='regression',
        **common)))
    return pose, affinity


def run_case(name, graph, cls, kwargs, task='classification', lr=2e-3, wd=1e-4, seed=2,
             sd_from=None, with_grads=True, with_adam=False, use_labels=False):
    torch.manual_seed(seed)
    np.random.seed(seed)
    kw = dict(kwargs)
    kw.setdefault('model_task', task)
    with tempfile.TemporaryDirectory() as tmp:
        model = cls(Path(tmp), lr, wd, None, None, silent=True, **kw)
    model = model.eval()  # no dropout anywhere; eval == train numerically, as in the ref tests
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}

    rec = {}

    def hook(idx):
        def fn(mod, args, kwargs_, out):
            h, coord, _, m = out
            rec[f'h{idx}'] = h.detach().clone()
            rec[f'x{idx}'] = coord.detach().clone()
            if m is not None:
                rec['m_last'] = m.detach().clone()
            if isinstance(mod, EGNNLayer):
                if mod.att_val is not None:
                    rec[f'att{idx}'] = np.array(mod.att_val, dtype=np.float32)
                if mod.node_att_val is not None:
                    rec[f'natt{idx}'] = np.array(mod.node_att_val, dtype=np.float32)
        return fn

    handles = [layer.register_forward_hook(hook(i), with_kwargs=True)
               for i, layer in enumerate(model.layers)]
    g = clone_graph(graph)
    y_pred, y_true, _, _ = model.unpack_input_data_and_predict(g)
    for h_ in handles:
        h_.remove()
    if use_labels:
        y_true = y_true.reshape(-1).float()   # the labels the loader read from the types file
    elif task == 'classification':
        y_true = torch.ones_like(y_pred)  # BCE against y=1
    else:
        y_true = torch.full_like(y_pred, 6.5)
    loss = model.get_loss(y_true, y_pred)
    model.optimiser.zero_grad()
    loss.backward()
    grads, grad_none = {}, []
    for pname, p in model.named_parameters():
        if p.grad is None:
            grad_none.append(pname)
        else:
            grads[pname] = p.grad.detach().clone()

    # one reference optimiser step through the reference's own backprop() (clip 1.0 + Adam)
    g = clone_graph(graph)
    y_pred2, _, _, _ = model.unpack_input_data_and_predict(g)
    model.backprop(y_true, y_pred2)
    sd1 = {k: v.detach().clone() for k, v in model.state_dict().items()}

    out = {
        'cfg': np.array(json.dumps({
            'case': name, 'class': cls.__name__, 'kwargs': kw, 'task': task, 'lr': lr, 'wd': wd,
            'seed': seed, 'grad_none': grad_none, 'sd_from': sd_from,
            'param_order': [n for n, _ in model.named_parameters()]})),
        'in/x': graph.x.numpy().astype(np.float32),
        'in/pos': graph.pos.numpy().astype(np.float32),
        'in/edge_index': graph.edge_index.numpy().astype(np.int32),
        'in/edge_type': graph.edge_attr.argmax(1).numpy().astype(np.uint8),
        'in/batch': graph.batch.numpy().astype(np.int32),
        'in/y_true': y_true.detach().numpy().astype(np.float32),
        'out/logits': y_pred.detach().numpy().astype(np.float32),
        'out/loss': np.float32(loss.item()),
    }
    assert bool((graph.edge_attr.sum(1) == 1).all())
    for k, v in rec.items():
        if k == 'm_last':  # [E,H] is the bulk of a fixture: keep sums that pin every element
            m = v.numpy().astype(np.float64)
            out['out/m_rowsum'] = m.sum(1).astype(np.float32)
            out['out/m_colsum'] = m.sum(0).astype(np.float32)
            out['out/m_rows16'] = v.numpy()[::16].astype(np.float32)
            continue
        out[f'out/{k}'] = v.numpy().astype(np.float32) if torch.is_tensor(v) else v
    if sd_from is None:
        for k, v in sd0.items():
            out[f'sd/{k}'] = v.numpy()
    else:  # same class, kwargs and seed => bit-identical init; keep one copy only
        other = np.load(HERE / f'{sd_from}.npz')
        for k, v in sd0.items():
            assert np.array_equal(other[f'sd/{k}'], v.numpy()), (name, k)
    if with_grads:
        for k, v in grads.items():
            out[f'grad/{k}'] = v.numpy()
    if with_adam:
        for k, v in sd1.items():
            out[f'adam/{k}'] = v.numpy()
    path = HERE / f'{name}.npz'
    np.savez_compressed(path, **out)
    print(f'{name:34s} N={graph.x.shape[0]:4d} E={graph.edge_index.shape[1]:5d} '
          f'logit={y_pred.detach().numpy().ravel()[:2]} loss={loss.item():.6f} '
          f'none={len(grad_none)} {path.stat().st_size / 1024:.0f} KiB')


def main():
    g1, g2, g3 = reference_test_graphs()
    g4 = Batch.from_data_list([synthetic_ball_graph(64, 8, 4.0, seed=11)])
    g5 = Batch.from_data_list([synthetic_ball_graph(48, 6, 5.0, seed=12),
                               synthetic_ball_graph(40, 5, 5.0, seed=13),
                               synthetic_ball_graph(56, 7, 5.0, seed=14)])
    g5.y = torch.tensor([1, 0, 1])

    test_kwargs = {  # test/setup_and_params.py:72-87
        'cache': False, 'k': 32, 'num_layers': 6, 'dropout': 0, 'dim_input': 12,
        'dim_output': 1, 'dim_hidden': 32, 'pooling_only': True, 'graphnorm': True,
        'update_coords': True, 'node_attention': True, 'residual': True,
        'edge_attention': True, 'softmax_attention': True}
    cli_default = {  # parse_args.py store_true flags all False, point_vs.py:189-221
        'dim_input': 12, 'k': 32, 'dim_output': 1, 'num_layers': 3, 'residual': False,
        'edge_residual': False, 'edge_attention': False, 'normalize': False, 'tanh': False,
        'dropout': 0.0, 'graphnorm': False, 'update_coords': True,
        'permutation_invariance': False, 'node_attention': False, 'gated_residual': False,
        'rezero': False, 'softmax_attention': False}

    def var(**changes):
        kw = dict(cli_default, num_layers=2, k=16)
        kw.update(changes)
        return kw

    # `--add`: keep the committed vectors and write only the cases that do not exist yet
    add_only = '--add' in sys.argv[1:]
    if not add_only:
        for f in sorted(HERE.glob('c[0-9]_*.npz')):
            f.unlink()
    _run_case = globals()['run_case']

    def run_case(name, *a, **k):
        if add_only and (HERE / f'{name}.npz').exists():
            return
        _run_case(name, *a, **k)
    # C1: the reference tests' own model kwargs on its own graphs
    run_case('c1_testkwargs_g1', g1, SartorrasEGNN, test_kwargs, with_adam=True)
    run_case('c1_testkwargs_g2', g2, SartorrasEGNN, test_kwargs, sd_from='c1_testkwargs_g1')
    run_case('c1_testkwargs_g3rot', g3, SartorrasEGNN, test_kwargs,
             sd_from='c1_testkwargs_g1', with_grads=False)
    # C0: CLI-default flag set (BASELINE config 2 shape at fixture size)
    run_case('c0_clidefault_g1', g1, SartorrasEGNN, cli_default)
    run_case('c0_clidefault_g4dup', g4, SartorrasEGNN, cli_default, sd_from='c0_clidefault_g1')
    run_case('c0_clidefault_g5batch', g5, SartorrasEGNN, cli_default,
             sd_from='c0_clidefault_g1', with_adam=True)
    run_case('c0_multitask_cls_g5batch', g5, MultitaskSatorrasEGNN, cli_default, with_adam=True)
    run_case('c0_multitask_reg_g5batch', g5, MultitaskSatorrasEGNN, cli_default,
             task='regression', sd_from='c0_multitask_cls_g5batch')
    # C2: BASELINE config 3 flag set (sigmoid edge gate + node gate), k=64 and k=32
    run_case('c2_sigatt_k64_g4dup', g4, SartorrasEGNN,
             var(k=64, edge_attention=True, node_attention=True))
    run_case('c2_sigatt_k32_g5batch', g5, SartorrasEGNN,
             var(k=32, edge_attention=True, node_attention=True, residual=True),
             with_adam=True)
    # C3: every remaining flag, mostly singly
    run_case('c3_normalize_tanh_g4', g4, SartorrasEGNN, var(normalize=True, tanh=True))
    run_case('c3_normalize_g5', g5, SartorrasEGNN, var(normalize=True))
    run_case('c3_tanh_k32_g4', g4, SartorrasEGNN, var(tanh=True, k=32))
    run_case('c3_residual_g4', g4, SartorrasEGNN, var(residual=True))
    run_case('c3_rezero_g4', g4, SartorrasEGNN, var(residual=True, rezero=True))
    run_case('c3_gated_g4', g4, SartorrasEGNN, var(residual=True, gated_residual=True))
    run_case('c3_edgeres_g4', g4, SartorrasEGNN, var(edge_residual=True, num_layers=3))
    run_case('c3_edgeres_gated_g5', g5, SartorrasEGNN,
             var(edge_residual=True, residual=True, gated_residual=True, num_layers=3))
    run_case('c3_edgeres_rezero_g4', g4, SartorrasEGNN,
             var(edge_residual=True, residual=True, rezero=True, num_layers=3))
    run_case('c3_perminv_g4', g4, SartorrasEGNN, var(permutation_invariance=True))
    run_case('c3_nocoords_g4', g4, SartorrasEGNN, var(update_coords=False))
    run_case('c3_graphnorm_g5', g5, SartorrasEGNN, var(graphnorm=True, residual=True))
    run_case('c3_att_tanh_g4', g4, SartorrasEGNN,
             var(edge_attention=True, node_attention=True, attention_activation_fn='tanh'))
    run_case('c3_att_relu_g4', g4, SartorrasEGNN,
             var(edge_attention=True, node_attention=True, attention_activation_fn='relu'))
    run_case('c3_att_silu_g4', g4, SartorrasEGNN,
             var(edge_attention=True, node_attention=True, attention_activation_fn='silu'))
    run_case('c3_softmax_g5', g5, SartorrasEGNN,
             var(edge_attention=True, softmax_attention=True, node_attention=True))
    run_case('c3_multifc_softplus_g4', g4, SartorrasEGNN,
             var(multi_fc=True, final_softplus=True), task='regression')
    run_case('c3_all_on_k32_g5', g5, SartorrasEGNN,
             var(k=32, num_layers=3, residual=True, gated_residual=True, edge_residual=True,
                 edge_attention=True, node_attention=True, normalize=True, tanh=True,
                 graphnorm=True))
    # C4: k=64 (the two-wave team kernels of the backward) with and without edge residual / gates
    run_case('c4_k64_clidefault_g4', g4, SartorrasEGNN, var(k=64))
    run_case('c4_k64_edgeres_g5', g5, SartorrasEGNN, var(k=64, edge_residual=True, num_layers=3))
    run_case('c4_k64_edgeres_rezero_att_g4', g4, SartorrasEGNN,
             var(k=64, edge_residual=True, residual=True, rezero=True, edge_attention=True,
                 node_attention=True, num_layers=3))
    run_case('c4_k64_normalize_tanh_att_g5', g5, SartorrasEGNN,
             var(k=64, normalize=True, tanh=True, edge_attention=True, residual=True), with_adam=True)
    run_case('c4_k64_softmax_g5', g5, SartorrasEGNN,
             var(k=64, edge_attention=True, softmax_attention=True, node_attention=True))
    # C5: BASELINE config 1 = the README example: `multitask ... --model_task both --layers 3` on the
    # reference's real data through its own loader, every model kwarg as point_vs.py:189-221 sets it
    pose, affinity = config1_graphs()
    readme_kwargs = {
        'act': 'relu', 'bn': True, 'cache': False, 'ds_frac': 1.0, 'k': 32, 'num_layers': 3,
        'dropout': 0.0, 'dim_input': 22, 'dim_output': 1, 'norm_coords': False,
        'norm_feats': False, 'thin_mlps': False, 'edge_attention': False, 'attention': False,
        'tanh': False, 'normalize': False, 'residual': False, 'edge_residual': False,
        'graphnorm': False, 'multi_fc': False, 'update_coords': True, 'node_final_act': False,
        'permutation_invariance': False, 'attention_activation_fn': 'sigmoid',
        'node_attention': False, 'gated_residual': False, 'rezero': False,
        'include_strain_info': False, 'final_softplus': False, 'softmax_attention': False}
    run_case('c5_config1_pose_real3', pose, MultitaskSatorrasEGNN, readme_kwargs,
             use_labels=True, with_adam=True)
    run_case('c5_config1_affinity_real3', affinity, MultitaskSatorrasEGNN, readme_kwargs,
             task='regression', use_labels=True, sd_from='c5_config1_pose_real3')
    run_case('c4_k32_softmax_edgeres_g4', g4, SartorrasEGNN,
             var(k=32, edge_attention=True, softmax_attention=True, edge_residual=True, residual=True,
                 num_layers=3))
    # C6: MultitaskSatorrasEGNN's per-layer attention placement (egnn_multitask.py:99-123): the edge gate
    # only in the LAST layer, the node gate only in the FIRST; and the mirrored placement
    run_case('c6_multitask_att_placement_g5', g5, MultitaskSatorrasEGNN,
             var(k=32, num_layers=3, edge_attention=True, node_attention=True, residual=True,
                 edge_attention_final_only=True, node_attention_first_only=True), with_adam=True)
    run_case('c6_multitask_att_placement_mirrored_g4', g4, MultitaskSatorrasEGNN,
             var(k=64, num_layers=3, edge_attention=True, node_attention=True,
                 edge_attention_first_only=True, node_attention_final_only=True), task='regression')
    # C7 (round 6): the edge-residual kinds WITHOUT edge attention at 32 and 64 channels on a graph of several hundred
    # 32-edge tiles. Until round 6 the only gated-edge-residual cases were c3_edgeres_gated_g5 (16 channels: the generic
    # kernels) and c3_all_on_k32_g5 (with attention), and every fixture graph was a few tiles - the H = 32 backward for
    # gated edge residual without attention was wrong on larger graphs and no case could see it
    # (profiles/r06_gated_residual_backward_defect.txt).
    g6 = Batch.from_data_list([synthetic_ball_graph(400, 20, 6.0, seed=21)])
    run_case('c7_k32_edgeres_gated_g6', g6, SartorrasEGNN,
             var(k=32, edge_residual=True, residual=True, gated_residual=True, num_layers=3), with_adam=True)
    run_case('c7_k32_edgeres_rezero_g6', g6, SartorrasEGNN,
             var(k=32, edge_residual=True, residual=True, rezero=True, num_layers=3))
    run_case('c7_k32_edgeres_sum_tanh_g6', g6, SartorrasEGNN,
             var(k=32, edge_residual=True, tanh=True, normalize=True, num_layers=3))
    run_case('c7_k64_edgeres_gated_g6', g6, SartorrasEGNN,
             var(k=64, edge_residual=True, residual=True, gated_residual=True, num_layers=3))
    # ... and the attention instantiations on the same graph (until round 6 only on fixture graphs of a few tiles)
    run_case('c7_k32_sigatt_residual_g6', g6, SartorrasEGNN,
             var(k=32, edge_attention=True, node_attention=True, residual=True, num_layers=3))
    run_case('c7_k32_softmax_edgeres_rezero_g6', g6, SartorrasEGNN,
             var(k=32, edge_attention=True, softmax_attention=True, edge_residual=True, residual=True, rezero=True,
                 normalize=True, num_layers=3))
    run_case('c7_k64_tanhatt_edgeres_gated_g6', g6, SartorrasEGNN,
             var(k=64, edge_attention=True, node_attention=True, attention_activation_fn='tanh', edge_residual=True,
                 residual=True, gated_residual=True, tanh=True, num_layers=3))


if __name__ == '__main__':
    main()
