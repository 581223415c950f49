#!/usr/bin/env python3
"""Reference-generated TRAINING TRAJECTORIES for the harness row.

Build container only (needs /root/reference; same import-only stand-ins as make_golden.py). Drives the
reference's own `train_model` (point_neural_network_base.py:136-205, backprop :417-429, training_setup
:372-388, on_epoch_end :470-490, save :501-517) on a FIXED list of batches (the weighted sampler is
bypassed: a plain list of collated batches stands in for the DataLoader) and records, per schedule:

    default           SartorrasEGNN, Adam lr 2e-3 wd 1e-4, 3 epochs x 4 batches
    one_cycle         the same with use_1cycle=True        (OneCycleLR over all 12 steps)
    warm_restarts     the same with warm_restarts=True     (CosineAnnealingWarmRestarts, T_0 = 4)
    multitask         MultitaskSatorrasEGNN: set_task('classification'), 2 epochs x 4 pose batches, then
                      set_task('regression'), 1 epoch x 4 affinity batches (point_vs.py:258-270)
    k64_attention     BASELINE config 3's flag set at 64 channels, 3 layers, warm restarts

per step: the loss `backprop()` returned and the learning rate the step ran at; at the end: the
`state_dict`, the epoch counters, the checkpoint file names and the keys of a checkpoint dict.
Output: tests/golden/train_<schedule>.npz (data only).

Round 6: for `default` and `k64_attention` the reference's OWN last checkpoint
file is kept byte for byte (tests/golden/ckpt_<schedule>.pt: what its save() wrote, :501-517 - tensors, floats and
ints in dicts only), and a RESUMED reference run is recorded beside it: a fresh reference model, `load_weights` of that
file (:528-565), its eval-mode logits on the first batch (`resume/logits`), then ONE more `backprop()` on that batch -
the run's 13th optimiser step - with its loss, learning rate and the weights it leaves (`resume/loss`, `resume/lr`,
`sd13/*`). For `k64_attention` (edge + node attention) a LEGACY-NAMED variant of the checkpoint is written too
(ckpt_k64_attention_legacy.pt): the names `_transform_names` (:519-526) maps FROM - `edge_attention_mlp` with its Linear
at index 2 of the older four-element Sequential, `node_attention_mlp` - i.e. the inverse of the reference's own
renaming, and it is only kept after the reference's `load_weights` has loaded it through its fallbacks and reproduced
the plain checkpoint's logits bit for bit.
"""
import json
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden as mg  # noqa: E402  (sets up the stubs, sys.path and cwd; defines the graph builders)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from point_vs.models.geometric.egnn_multitask import MultitaskSatorrasEGNN  # noqa: E402
from point_vs.models.geometric.egnn_satorras import SartorrasEGNN  # noqa: E402
from torch_geometric.data import Batch  # noqa: E402  (stub)


class ListLoader(list):
    """What train_model needs of a DataLoader: len(), iteration, .batch_size. Every pass yields fresh copies, as a
    DataLoader does: the reference's layers update `graph.pos` IN PLACE, which ties a batch object to the
    autograd graph of the step that used it."""
    batch_size = 3

    def __iter__(self):
        for b in list.__iter__(self):
            yield mg.clone_graph(b)


def batches(seed0, regression=False):
    out = ListLoader()
    for b in range(4):
        items = [mg.synthetic_ball_graph(n, nl, 5.0, seed=seed0 + 10 * b + k)
                 for k, (n, nl) in enumerate(((44, 6), (36, 5), (52, 7)))]
        batch = Batch.from_data_list(items)
        if regression:
            batch.y = torch.tensor([6.5, 4.75, 8.0]) + 0.25 * b
        else:
            batch.y = torch.tensor([1, 0, 1]) if b % 2 == 0 else torch.tensor([0, 1, 0])
        out.append(batch)
    return out


KW = {'dim_input': 12, 'k': 32, 'dim_output': 1, 'num_layers': 2, 'residual': True, 'edge_residual': False,
      'edge_attention': True, 'normalize': False, 'tanh': True, 'dropout': 0.0, 'graphnorm': False,
      'update_coords': True, 'permutation_invariance': False, 'node_attention': False, 'gated_residual': False,
      'rezero': False, 'softmax_attention': False, 'model_task': 'classification'}


def record_training(model, phases):
    """phases: [(task, loader, epochs)]. Returns the per-step records."""
    steps = []
    real_backprop = model.backprop

    def backprop(y_true, y_pred):
        lr = model.optimiser.param_groups[0]['lr']
        loss = real_backprop(y_true, y_pred)
        steps.append((model.model_task, float(loss), float(lr)))
        return loss
    model.backprop = backprop
    model.eta = '0'                      # (train_model reads it for the wandb record)
    for task, loader, epochs in phases:
        model.set_task(task)
        model.train_model(loader, epochs=epochs)
    return steps


def resumed_reference_run(name, cls, ctor_kwargs, kw, run_dir, ckpt_rel, loader, seed):
    """Keeps the reference-written checkpoint `run_dir / ckpt_rel` as tests/golden/ckpt_<name>.pt and records what the
    REFERENCE does when it resumes from it (module docstring). Returns the arrays to add to the schedule's file."""
    import shutil
    src = run_dir / ckpt_rel
    kept = HERE / f'ckpt_{name}.pt'
    shutil.copyfile(src, kept)
    first = next(iter(list.__iter__(loader)))

    def fresh():
        torch.manual_seed(seed + 1000)          # other initial weights than the run's: everything must come from the file
        np.random.seed(seed + 1000)
        tmp2 = Path(tempfile.mkdtemp())
        return cls(tmp2, 2e-3, 1e-4, None, None, silent=True, **ctor_kwargs, **kw), tmp2

    def place(tmp2, path):
        # load_weights reads model_kwargs.yaml two levels above the file (:534-536); a model built with silent=True
        # does not write one (:109-112), so the same dump is made here
        import yaml
        dst = tmp2 / 'checkpoints' / Path(ckpt_rel).name
        dst.parent.mkdir(parents=True, exist_ok=True)
        (tmp2 / 'model_kwargs.yaml').write_text(yaml.dump(dict(kw)))
        shutil.copyfile(path, dst)
        return dst

    def logits_of(model):
        model.eval()
        with torch.no_grad():
            y = model(mg.clone_graph(first))
        return y.detach().reshape(-1).numpy().astype(np.float32)

    model, tmp2 = fresh()
    model.load_weights(place(tmp2, kept), silent=True)
    out = {'resume/logits': logits_of(model),
           'resume/p_epoch': np.array(model.p_epoch), 'resume/a_epoch': np.array(model.a_epoch)}
    model.train()
    model.eta = '0'
    y_pred, y_true, _, _ = model.unpack_input_data_and_predict(mg.clone_graph(first))
    out['resume/lr'] = np.array(float(model.optimiser.param_groups[0]['lr']))
    out['resume/loss'] = np.array(float(model.backprop(y_true, y_pred)))
    for k, v in model.state_dict().items():
        out[f'sd13/{k}'] = v.detach().numpy().copy()
    st = model.optimiser.state_dict()['state']
    out['resume/optimiser_steps'] = np.array(sorted({int(s['step']) for s in st.values()}))
    print(f'{name:14s} resumed: logits {out["resume/logits"]}  13th-step loss {float(out["resume/loss"]):.6f} '
          f'lr {float(out["resume/lr"]):.3e}  {kept.name} {kept.stat().st_size / 1024:.0f} KiB')

    if kw.get('edge_attention') and kw.get('node_attention'):
        ck = torch.load(kept, map_location='cpu', weights_only=False)
        import re
        from collections import OrderedDict
        legacy = OrderedDict()
        for k, v in ck['model_state_dict'].items():      # the inverse of _transform_names + the older Sequential
            k = re.sub(r'(^|\.)att_mlp\.0\.', r'\1edge_attention_mlp.2.', k)
            k = k.replace('node_att_mlp', 'node_attention_mlp')
            legacy[k] = v
        assert any('edge_attention_mlp.2.' in k for k in legacy) and any('node_attention_mlp' in k for k in legacy)
        ck['model_state_dict'] = legacy
        legacy_path = HERE / f'ckpt_{name}_legacy.pt'
        torch.save(ck, legacy_path)
        model2, tmp3 = fresh()
        model2.load_weights(place(tmp3, legacy_path), silent=True)     # the reference loads it (through both fallbacks)
        again = logits_of(model2)
        assert np.array_equal(again, out['resume/logits']), (again, out['resume/logits'])
        print(f'{name:14s} legacy-named variant loaded by the reference: logits identical  {legacy_path.name}')
    return out


def run(name, cls, ctor_kwargs, phases, seed=7, kw_changes=None, keep_checkpoint=False):
    torch.manual_seed(seed)
    np.random.seed(seed)
    KW = dict(globals()['KW'], **(kw_changes or {}))
    with tempfile.TemporaryDirectory() as tmp:
        model = cls(Path(tmp), 2e-3, 1e-4, None, None, silent=True, **ctor_kwargs, **KW)
        sd0 = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
        steps = record_training(model, phases)
        ckpts = sorted(str(p.relative_to(tmp)) for p in Path(tmp).rglob('*.pt'))
        ck = torch.load(Path(tmp) / ckpts[-1], map_location='cpu', weights_only=False)
        ckpt_keys = sorted(ck.keys())
        opt_steps = sorted({int(s['step']) for s in ck['optimiser_state_dict']['state'].values()})
        resume = None
        if keep_checkpoint:
            resume = resumed_reference_run(name, cls, ctor_kwargs, KW, Path(tmp), ckpts[-1], phases[-1][1], seed)
    out = {'meta': np.array(json.dumps({
        'name': name, 'class': cls.__name__, 'kwargs': KW, 'ctor': ctor_kwargs, 'seed': seed, 'lr': 2e-3, 'wd': 1e-4,
        'phases': [(t, len(l), e) for t, l, e in phases], 'tasks': [s[0] for s in steps],
        'p_epoch': model.p_epoch, 'a_epoch': model.a_epoch, 'global_iter': model.global_iter,
        'checkpoints': ckpts, 'checkpoint_keys': ckpt_keys, 'optimiser_steps_in_last_checkpoint': opt_steps})),
        'loss': np.array([s[1] for s in steps], dtype=np.float64),
        'lr': np.array([s[2] for s in steps], dtype=np.float64)}
    if resume is not None:
        out.update(resume)
    for k, v in sd0.items():
        out[f'sd0/{k}'] = v
    for k, v in model.state_dict().items():
        out[f'sd1/{k}'] = v.detach().numpy()
    # the batches (inputs): every loader's batches, in order
    for pi, (_, loader, _) in enumerate(phases):
        for bi, b in enumerate(list.__iter__(loader)):
            pre = f'in/p{pi}b{bi}/'
            out[pre + 'x'] = b.x.numpy().astype(np.float32)
            out[pre + 'pos'] = b.pos.numpy().astype(np.float32)
            out[pre + 'edge_index'] = b.edge_index.numpy().astype(np.int32)
            out[pre + 'edge_type'] = b.edge_attr.argmax(1).numpy().astype(np.uint8)
            out[pre + 'batch'] = b.batch.numpy().astype(np.int32)
            out[pre + 'y'] = b.y.numpy().astype(np.float32)
    path = HERE / f'train_{name}.npz'
    np.savez_compressed(path, **out)
    print(f'{name:14s} steps={len(steps)} loss {steps[0][1]:.6f} -> {steps[-1][1]:.6f}  lr {steps[0][2]:.3e} .. '
          f'{max(s[2] for s in steps):.3e}  ckpts={ckpts}  {path.stat().st_size / 1024:.0f} KiB')


def main():
    pose = batches(100)
    run('default', SartorrasEGNN, {}, [('classification', pose, 3)], keep_checkpoint=True)
    run('one_cycle', SartorrasEGNN, {'use_1cycle': True}, [('classification', batches(100), 3)])
    run('warm_restarts', SartorrasEGNN, {'warm_restarts': True}, [('classification', batches(100), 3)])
    run('multitask', MultitaskSatorrasEGNN, {},
        [('classification', batches(100), 2), ('regression', batches(500, regression=True), 1)])
    # BASELINE config 3's flag set (64 channels, sigmoid edge gate + node gate): the H = 64 kernels in a training run
    run('k64_attention', SartorrasEGNN, {'warm_restarts': True}, [('classification', batches(100), 3)],
        kw_changes={'k': 64, 'edge_attention': True, 'node_attention': True, 'tanh': False, 'num_layers': 3},
        keep_checkpoint=True)


if __name__ == '__main__':
    main()
